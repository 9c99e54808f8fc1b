"""Build check on the CPU (hipcc cross-compiles gfx950): no kernel of the library keeps registers in scratch memory.

Why a test: scratch is invisible in the source and expensive on this path -- round 2's ``k_grad`` spilled 260 registers per lane in its
epilogue (4.4 GB of HBM writes per launch at C2), and a ``noinline`` device function in the diagonal kernel saved / restored 112 VGPRs through
scratch on every call, which the resource-usage remark of the calling KERNEL did not even show. So every
function the compiler reports for every source -- kernels and any out-of-line device function -- must have ScratchSize 0 and no VGPR spill."""
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / 'rom-comma_amd' / 'csrc'
SOURCES = ('api', 'gemm', 'gram', 'potrf', 'solve', 'sobol')


def _resource_report(name: str):
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '--offload-arch=gfx950', '-Wno-unused-value', '-c',
           f'{name}.hip', '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
    done = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    functions, current = {}, None
    for line in done.stderr.splitlines():
        if (m := re.search(r'Function Name: (\S+)', line)):
            current = functions.setdefault(m.group(1), {})
        elif current is not None and (m := re.search(r'(ScratchSize \[bytes/lane\]|VGPRs Spill|VGPRs): (\d+)', line)):
            current[m.group(1)] = int(m.group(2))
    return name, functions


def test_no_kernel_uses_scratch_memory():
    with ThreadPoolExecutor(max_workers=6) as pool:
        reports = dict(pool.map(_resource_report, SOURCES))
    kernels = {f'{src}:{fn}': usage for src, functions in reports.items() for fn, usage in functions.items()}
    assert len(kernels) >= 40, sorted(kernels)                                   # the remark pass really ran on every source
    offenders = {k: u for k, u in kernels.items() if u.get('ScratchSize [bytes/lane]', 0) != 0 or u.get('VGPRs Spill', 0) != 0}
    assert not offenders, offenders
    assert any('k_grad' in k for k in kernels) and any('k_diag_factor' in k for k in kernels) and any('k_trsm_subst' in k for k in kernels)
    # two workgroups of the GEMM family per CU (four waves per SIMD) need <= 128 VGPRs: the kernels that run beside each other in the factorisation
    for needle in ('k_syrk_lower', 'k_gemm_nt_subILi4ELi3E', 'k_trsm_subst', 'k_gradILi33', 'k_trtri_T', 'k_trtri_X'):
        hits = [u for k, u in kernels.items() if needle in k]
        assert hits and all(u['VGPRs'] <= 128 for u in hits), (needle, hits)
