from bonito_amd.crf.basecall import *  # noqa: F401,F403  (same pipeline, reference transformer/basecall.py:1)
from bonito_amd.crf.basecall import basecall  # noqa: F401
