"""CSV-backed parameter store: Frame, Data, Model (the "Model/Store" half of the plugin API)."""
from romcomma_amd.base.definitions import *        # noqa: F401,F403
from romcomma_amd.base.classes import Frame, Data, Model   # noqa: F401
