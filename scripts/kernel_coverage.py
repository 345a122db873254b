"""Kernel-instantiation coverage of the GPU test suite.
On the GPU box:
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cov -- python -m pytest $GRAFT_REPO_ROOT/tests -m gpu -q
    python scripts/kernel_coverage.py gpurun_out/cov [more directories] > profiles/r06_kernel_coverage.txt
Every `*kernel_stats.csv` / `*kernel_trace.csv` under the directory is read (the suite's child processes write their own),
kernel names are normalised like scripts/kernel_symbols.py does, and the list of the library's kernels is printed with the number
of launches the suite made of each; the ones never launched come last under `UNLAUNCHED`.  Exit code 1 if any is unlaunched."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_symbols as ks  # noqa: E402


def launched(directory):
    counts = {}
    files = glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True)
    use_stats = bool(files)
    if not files:
        files = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name") or row.get("Kernel_Name") or ""
                n = int(row.get("Calls", 1)) if use_stats else 1
                key = ks.normalize(name)
                counts[key] = counts.get(key, 0) + n
    return counts, len(files)


def main():
    dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    lib = libs[0] if libs else ks.DEFAULT_LIB
    counts, nfiles = {}, 0
    for d in dirs:                       # several runs (the suite in pieces): launches add up
        c, nf = launched(d)
        nfiles += nf
        for k, v in c.items():
            counts[k] = counts.get(k, 0) + v
    mine = ks.kernels(lib)
    missing = [k for k in mine if counts.get(k, 0) == 0]
    print(f"# kernel-instantiation coverage of `pytest tests -m gpu` ({nfiles} rocprofv3 kernel-stats files, one per process)")
    print(f"# {len(mine)} kernels in {os.path.relpath(lib, ks.ROOT)}; {len(mine) - len(missing)} launched; {len(missing)} never launched")
    print("# launches  kernel")
    for k in mine:
        if counts.get(k, 0):
            print(f"{counts[k]:10d}  {k}")
    print("UNLAUNCHED")
    for k in missing:
        print(f"         0  {k}")
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
