"""Experiment: do single-wave workgroups without LDS run BESIDE the sliced product (one 15-wave, 160-KiB workgroup per CU)?
A spin kernel (scripts/exp_spin.hip -> scripts/bin/libspin.so: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/exp_spin.hip -o scripts/bin/libspin.so;
256..1024 workgroups of 64 threads, pure VALU) on a second stream while the product
runs on the first.   python scripts/exp_coresident.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
from bench import make_graph

dev = torch.device("cuda:0")
n, C = 132534, 64
ei = make_graph(n, 39561252, dev)
be = ops.get_backend()
x = torch.randn(n, C, device=dev)
csr = ops.csr_cache.get(ei, None, n, C * 4)
sl = csr.sliced(0, n, C)
ys = be.sliced_prescale(x, csr.rowptr, n, sl.plan)
spmm = lambda: be.sliced_spmm(sl, ys, csr.rowptr, n, 0, n, C, None, 1.0, 1.0)
spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libspin.so"))
spin.spin_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
buf = torch.zeros(4096, device=dev)
side = torch.cuda.Stream(dev)


def timed(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


def spin_on(stream, blocks, iters):
    spin.spin_launch(buf.data_ptr(), blocks, iters, ctypes.c_void_p(stream.cuda_stream))


main = torch.cuda.current_stream(dev)
t_spmm = timed(spmm)
for blocks, iters in ((256, 20000), (512, 20000), (1024, 10000), (256, 60000)):
    t_spin = timed(lambda: spin_on(main, blocks, iters))

    def both():
        side.wait_stream(main)
        spmm()
        with torch.cuda.stream(side):
            spin_on(side, blocks, iters)
        main.wait_stream(side)

    def both_spin_first():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            spin_on(side, blocks, iters)
        spmm()
        main.wait_stream(side)
    print(f"product alone {t_spmm:.0f} us, spin ({blocks} x 64 threads, {iters} iterations) alone {t_spin:.0f} us, "
          f"together {timed(both):.0f} us (product enqueued first) / {timed(both_spin_first):.0f} us (spin first)", flush=True)
