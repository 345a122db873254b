"""Where a spatial-temporal training snapshot spends its host time: cProfile of cumulative epochs at chickenpox size."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from st_epoch import data, epoch
from difformer_amd import DIFFormer

kernel = sys.argv[1] if len(sys.argv) > 1 else "simple"
n, d, deg, T = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (20, 4, 4, 104)
cumulative = (sys.argv[6] != "inc") if len(sys.argv) > 6 else True
dev = torch.device("cuda:0")
host = data(n, d, deg, T, False, False)
model = DIFFormer(d, 4, 1, num_layers=2, alpha=0.5, dropout=0.2, num_heads=1, kernel=kernel, use_bn=True, use_residual=True,
                  use_graph=True, use_weight=False).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=0.01)
fresh = lambda: [tuple(a.clone().to(dev) for a in s) for s in host]
for _ in range(2):
    epoch(model, fresh(), opt, cumulative)
snaps = fresh()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
epoch(model, snaps, opt, cumulative)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
