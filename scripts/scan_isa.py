"""Per-kernel view of the generated gfx950 code: loads that are waited for right where they are issued (a load directly followed
by `s_waitcnt vmcnt(0)`: a serialised memory round trip), basic blocks, scratch (spill) and flat instructions.
    python scripts/scan_isa.py [substring of a kernel name ...]        (no GPU needed; compiles csrc/*.hip to assembly under /tmp)
With names: also prints, per basic block, the order of loads (L), flat loads (F), waits (|n|), LDS reads (d), MFMAs (M), scratch (S).
profiles/r05_experiments.md section 12 lists what this found."""
import glob, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "difformer_amd", "csrc")
OUT = "/tmp/dif_isa"
os.makedirs(OUT, exist_ok=True)
hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
procs = []
for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
    s = os.path.join(OUT, os.path.basename(f)[:-4] + ".s")
    if not os.path.exists(s) or os.path.getmtime(s) < max(os.path.getmtime(f), *(os.path.getmtime(h) for h in glob.glob(os.path.join(SRC, "*.h")))):
        procs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                                       "-S", "--cuda-device-only", f, "-o", s], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
filt = shutil.which("c++filt")


def demangle(n):
    if not filt:
        return n
    return subprocess.run([filt, n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")


want = sys.argv[1:]
rows = []
for s in sorted(glob.glob(os.path.join(OUT, "*.s"))):
    t = open(s).read()
    for m in re.finditer(r"^(_Z\S+):\s*(;.*)?$", t, re.M):
        i = m.end()
        j = t.find(".end_amdhsa_kernel", i)
        if j < 0:
            continue
        name = demangle(m.group(1))
        blocks, cur = [], []
        for l in t[i:j].splitlines():
            l = l.strip()
            if re.match(r"^\.LBB", l):
                blocks.append(cur)
                cur = []
            elif l.startswith(("global_load", "buffer_load")):
                cur.append("L")
            elif l.startswith("flat_load"):
                cur.append("F")
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                cur.append("|" + re.search(r"vmcnt\((\d+)\)", l).group(1) + "|")
            elif l.startswith("ds_read"):
                cur.append("d")
            elif l.startswith("v_mfma"):
                cur.append("M")
            elif l.startswith("scratch_"):
                cur.append("S")
        blocks.append(cur)
        sq = "#".join("".join(b) for b in blocks)
        rows.append((len(re.findall(r"[LF]\|0\|", sq)), sq.count("L") + sq.count("F"), len(blocks), sq.count("S"), sq.count("F"),
                     os.path.basename(s)[:-2], name, blocks))
rows.sort(key=lambda r: -r[0])
print("serialised loads | loads | blocks | scratch | flat | file | kernel")
for ser, nl, nb, ns, nf, f, name, blocks in rows:
    if want and not any(w in name for w in want):
        continue
    if not want and ser < 8 and nf == 0:
        continue
    print(f"{ser:4d} {nl:4d} {nb:4d} {ns:4d} {nf:4d}  {f}  {name[:150]}")
    if want:
        for b in blocks:
            sq = "".join(b)
            if sq:
                print("        " + sq[:240])
