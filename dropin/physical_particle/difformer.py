"""Put this file next to `physical particle/parse.py` (which does `from difformer import DIFFormer_v2`, parse.py:3;
the reference ships the module under the un-importable name `difformer-v2.py`) with the repo root on PYTHONPATH."""
from difformer_amd.difformer_v2 import *  # noqa: F401,F403
from difformer_amd.difformer_v2 import DIFFormer_v2, TransConv, gcn_conv, make_batch, make_batch_mask, to_pad  # noqa: F401
