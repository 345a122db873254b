"""Shared helpers of the `spatial-temporal` tests: the fixtures of tests/golden/golden_st.npz (written by
tests/golden/make_golden_st.py from `spatial-temporal/difformer.py` itself) and the caller lines of
`spatial-temporal/main.py:91-120` restated with citations (the GPU box has no /root/reference)."""
import numpy as np
import torch

from conftest import load_golden, split_model_case

ST = load_golden("st")


def cases(prefix):
    return sorted(n for n in ST if n.startswith(prefix + "/"))


def dense_graph(n):
    """`--special_treat dense`, main.py:100-102."""
    row = torch.arange(0, n).unsqueeze(1).repeat(1, n)
    col = torch.arange(0, n).unsqueeze(0).repeat(n, 1)
    return torch.stack([row.reshape(-1), col.reshape(-1)], dim=0)


def cost_fn(y_hat, y):
    return torch.mean((y_hat - y) ** 2)                                  # main.py:107 ([n,1] against [n]: as written)


def build_model(cls, c, device=None):
    """The model `parse_method` builds for a spatial-temporal command line (parse.py:54-55) with the fixture's parameters."""
    cfg, sd = split_model_case(c)
    model = cls(int(cfg["in_channels"]), int(cfg["hidden_channels"]), int(cfg["out_channels"]),
                num_layers=int(cfg["num_layers"]), alpha=float(cfg["alpha"]), dropout=0.0, num_heads=int(cfg["num_heads"]),
                kernel=str(cfg["kernel"]), use_bn=bool(cfg["use_bn"]), use_residual=bool(cfg["use_residual"]),
                use_graph=bool(cfg["use_graph"]), use_weight=bool(cfg["use_weight"]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    if device is not None:
        model = model.to(device)
    return model.train(), cfg                                            # main.py:91


def snapshots(c, device=None, fresh=True):
    """The T snapshots of a cumul/ case as (x, edge_index, edge_attr, y) tuples.  `fresh`: NEW tensor objects per snapshot
    even when the graph is static -- what `snapshot.to(device)` hands the model (main.py:96)."""
    T = c["x"].shape[0]
    out = []
    for t in range(T):
        ei = c[f"edge_index/{t}"] if f"edge_index/{t}" in c else c["edge_index/0"]
        tens = [torch.from_numpy(np.ascontiguousarray(a)) for a in (c["x"][t], ei, c[f"edge_weight/{t}"], c["y"][t])]
        if fresh:
            tens = [a.clone() for a in tens]
        if device is not None:
            tens = [a.to(device) for a in tens]
        out.append(tuple(tens))
    return out


def cumulative_epoch(model, snaps, optimizer=None):
    """main.py:86-120 for every dataset but wikimath: retain_grad on the parameters, the snapshots forwarded one after
    another, the costs SUMMED, ONE `cost_tr.backward(retain_graph=True)` -> (mean cost tensor, [y_hat per snapshot])."""
    for param in model.parameters():                                     # main.py:86-89
        if param.requires_grad:
            param.retain_grad()
    model.train()                                                        # main.py:91
    cost_tr = 0
    outs = []
    for time, (x, ei, ea, y) in enumerate(snaps):                        # main.py:94
        y_hat = model(x, ei, ea)                                         # main.py:105
        cost = cost_fn(y_hat, y)                                         # main.py:107
        cost_tr += cost                                                  # main.py:109
        outs.append(y_hat.detach())
    cost_tr = cost_tr / (time + 1)                                       # main.py:116
    cost_tr.backward(retain_graph=True)                                  # main.py:119
    if optimizer is not None:
        optimizer.step()                                                 # main.py:120-121
        optimizer.zero_grad()
    return cost_tr, outs


def incremental_epoch(model, snaps, optimizer):
    """main.py:110-114, the wikimath branch: backward and optimiser step per snapshot -> mean cost (float)."""
    model.train()
    cost_tr = 0
    for time, (x, ei, ea, y) in enumerate(snaps):
        y_hat = model(x, ei, ea)
        cost = cost_fn(y_hat, y)
        cost_tr += cost.detach().item()                                  # main.py:111
        cost.backward()
        optimizer.step()
        optimizer.zero_grad()
    return cost_tr / (time + 1)


@torch.no_grad()
def evaluate(model, snaps):
    """spatial-temporal/eval.py:5-23."""
    model.eval()
    cost_te = 0
    for time, (x, ei, ea, y) in enumerate(snaps):
        y_hat = model(x, ei, ea)
        cost_te += torch.mean((y_hat - y) ** 2)
    cost_te = cost_te / (time + 1)
    return cost_te.item()
