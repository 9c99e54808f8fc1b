"""Count the non-trivial lines (> 25 characters after stripping, comments and docstrings removed) of a file of this build that also occur
verbatim in a reference file -- the function-level copy check of VERDICT r2.  Build container only (needs /root/reference).

    python tools/verbatim_lines.py rom-comma_amd/data/storage.py /root/reference/romcomma/data/storage.py
"""
import io
import sys
import tokenize


def code_lines(path):
    src = open(path).read()
    drop = set()
    prev = None
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type == tokenize.COMMENT:
            drop.add((tok.start, tok.end))
        if tok.type == tokenize.STRING and (prev is None or prev.type in (tokenize.INDENT, tokenize.NEWLINE, tokenize.NL, tokenize.DEDENT)):
            drop.add((tok.start, tok.end))                      # a docstring / bare string statement
        if tok.type not in (tokenize.NL, tokenize.COMMENT):
            prev = tok
    lines = src.split('\n')
    for (r0, c0), (r1, c1) in sorted(drop, reverse=True):
        if r0 == r1:
            lines[r0 - 1] = lines[r0 - 1][:c0] + lines[r0 - 1][c1:]
        else:
            lines[r0 - 1] = lines[r0 - 1][:c0]
            for r in range(r0, r1 - 1):
                lines[r] = ''
            lines[r1 - 1] = lines[r1 - 1][c1:]
    return [ln.strip() for ln in lines if len(ln.strip()) > 25]


if __name__ == '__main__':
    mine, ref = code_lines(sys.argv[1]), set(code_lines(sys.argv[2]))
    same = [ln for ln in mine if ln in ref]
    print(f'{sys.argv[1]}: {len(same)} of {len(mine)} non-trivial lines verbatim in {sys.argv[2]} ({100.0 * len(same) / max(len(mine), 1):.1f} %)')
    for ln in same:
        print('   ', ln)
