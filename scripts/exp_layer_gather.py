"""Layer kernel with the aggregation inside (dif_simple_layer_gather_*) against SpMM launch + layer kernel, kernel times only.
    python scripts/exp_layer_gather.py            (GPU box; DIFFORMER_HIP_LIB picks a measurement build)
Random graph with `deg` incoming entries per row on average (Poisson) plus self loops; 64 columns; HIP events round 50
launches of each path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import ops  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    be = ops.get_backend()
    g = torch.Generator().manual_seed(0)
    print("lib", os.environ.get("DIFFORMER_HIP_LIB", "default"))
    for dtype in (torch.float32, torch.bfloat16):
        for n in (100000, 25000):
            for deg in (0, 2.3, 7):
                e = int(n * deg)
                ei = torch.cat([torch.randint(0, n, (2, e), generator=g), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
                csr = ops.csr_cache.get(ei, None, n, 64 * 4)
                x = torch.randn(n, 64, generator=g).to(dev).to(dtype)
                coef = torch.randn(64 * 64 + 64 + 64 + 4, generator=g).to(dev) * 0.1
                Wv, bv = (torch.randn(64, 64, generator=g) * 0.1).to(dev), torch.randn(64, generator=g).to(dev)
                lw, lb = torch.ones(64, device=dev), torch.zeros(64, device=dev)
                rs = csr.row_sums()

                def two():
                    ax = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, 0, n, None, 1.0, 1.0, None, None)
                    return be.simple_layer(x, coef, 64, ax, Wv, bv, rs, 1.0, None, True, 0.5, lw, lb, 1e-5, False)

                def layer_only(ax=be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, 0, n, None, 1.0, 1.0, None, None)):
                    return be.simple_layer(x, coef, 64, ax, Wv, bv, rs, 1.0, None, True, 0.5, lw, lb, 1e-5, False)

                def one():
                    return be.simple_layer(x, coef, 64, None, Wv, bv, rs, 1.0, None, True, 0.5, lw, lb, 1e-5, False,
                                           gather=(csr.rowptr, csr.src, csr.val))

                err = (one().float() - two().float()).abs().max().item()
                print(f"{str(dtype)[6:]:9s} n={n:6d} deg={deg + 1:4.1f}: spmm+layer {timed(two):6.1f} us (layer alone {timed(layer_only):5.1f})"
                      f"   gather-in-layer {timed(one):6.1f} us   max diff {err:.2e}")


def hubs():
    """One long row among short ones: the lock-step walk of the layer kernel waits for it (LAYER_GATHER_MAX_ROW)."""
    dev = torch.device("cuda:0")
    be = ops.get_backend()
    g = torch.Generator().manual_seed(1)
    for n in (100000, 2708):
        for hub in (0, 32, 64, 128, 512, 4096):
            e = int(n * 2.3)
            ei = torch.cat([torch.randint(0, n, (2, e), generator=g), torch.arange(n).repeat(2, 1),
                            torch.stack([torch.randint(0, n, (hub,), generator=g), torch.full((hub,), 7)])], dim=1).to(dev)
            csr = ops.csr_cache.get(ei, None, n, 64 * 4)
            x = torch.randn(n, 64, generator=g).to(dev)
            coef = torch.randn(64 * 64 + 64 + 64 + 4, generator=g).to(dev) * 0.1
            Wv, bv = (torch.randn(64, 64, generator=g) * 0.1).to(dev), torch.randn(64, generator=g).to(dev)
            lw, lb = torch.ones(64, device=dev), torch.zeros(64, device=dev)
            rs = csr.row_sums()

            def two():
                ax = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x, 0, n, None, 1.0, 1.0, None, None)
                return be.simple_layer(x, coef, 64, ax, Wv, bv, rs, 1.0, None, True, 0.5, lw, lb, 1e-5, False)

            def one():
                return be.simple_layer(x, coef, 64, None, Wv, bv, rs, 1.0, None, True, 0.5, lw, lb, 1e-5, False,
                                       gather=(csr.rowptr, csr.src, csr.val))

            print(f"n={n:6d} longest row {csr.max_degree():5d}: spmm+layer {timed(two):6.1f} us   gather-in-layer {timed(one):6.1f} us")


if __name__ == "__main__":
    hubs() if len(sys.argv) > 1 and sys.argv[1] == "hubs" else main()
