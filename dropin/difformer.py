"""Copy this file over `difformer.py` in a reference task folder (`node classification/`,
`image and text/`, `spatial-temporal/`) with the repo root on PYTHONPATH: the unchanged
`parse.py` (`from difformer import *`) then instantiates the MI355X implementation."""
from difformer_amd.difformer import *  # noqa: F401,F403
from difformer_amd.difformer import DIFFormer, DIFFormerConv, full_attention_conv, gcn_conv  # noqa: F401
