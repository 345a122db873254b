"""Properties of the generated gfx950 code that round 5 paid for (profiles/r05_experiments.md sections 11-12): the software-
pipelined kernels keep their loads in flight (no load waited for where it is issued, inside the sweep), spill nothing and use no
FLAT memory instructions.  hipcc cross-compiles without a GPU; the four sources compile side by side (~1 minute)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "difformer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FILES = ["sigmoid_attn", "sigmoid_attn_bwd", "simple_attn_bwd", "skinny_linear"]


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("isa")
    procs = {f: subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-S",
                                  "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", os.path.join(SRC, f + ".hip"), "-o",
                                  str(out / (f + ".s"))], stderr=subprocess.PIPE, text=True) for f in FILES}
    res = {}
    for f, p in procs.items():
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, err[-2000:]
        spills = {}
        name = None
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"VGPRs Spill: (\d+)", line)
            if m and name:
                spills[name] = int(m.group(1))
        res[f] = (open(out / (f + ".s")).read(), spills)
    return res


def kernels(text, needle):
    """{mangled name: [instruction mnemonic lines]} of the kernels whose mangled name contains `needle`."""
    found = {}
    for m in re.finditer(r"^(_Z\S+):\s*(;.*)?$", text, re.M):
        if needle not in m.group(1):
            continue
        j = text.find(".end_amdhsa_kernel", m.end())
        if j > 0:
            found[m.group(1)] = [l.strip() for l in text[m.end():j].splitlines()]
    return found


def shape(lines):
    """(self-waiting loads, flat instructions): a self-waiting load is a memory load directly followed by s_waitcnt vmcnt(0)."""
    seq = []
    for l in lines:
        if l.startswith(("global_load", "buffer_load", "flat_load")):
            seq.append("L")
        elif l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            seq.append("0")
        elif l.startswith(("v_", "s_", "ds_", "global_store", "scratch_")) and not l.startswith("s_nop"):
            seq.append("x")
    return len(re.findall("L0", "".join(seq))), sum(l.startswith(("flat_load", "flat_store")) for l in lines)


CASES = [
    # (file, substring of the mangled name, most self-waiting loads allowed)
    ("sigmoid_attn", "sigmoid_attn_kernelILb1ELb1ELb0EfLb1E", 2),                 # split-bf16 inference forward, aligned rows
    ("sigmoid_attn_bwd", "sigmoid_bwd_kernelILi0ELb1ELb0ELb0E", 2),               # fp32 dQ sweep
    ("sigmoid_attn_bwd", "sigmoid_bwd_kernelILi1ELb1ELb0ELb0E", 2),               # fp32 dK / dV sweep
    ("sigmoid_attn_bwd", "sigmoid_bwd_kernelILi1ELb1ELb1ELb0E", 2),               # ... of the batched (v2) attention
    ("simple_attn_bwd", "rowgemm_split_kernel", 4),
    ("simple_attn_bwd", "simple_bwd_prep_vec_kernelILi2E", 3),
    ("skinny_linear", "skinny_linear_bf16_kernelILi3E", 24),                      # the last (ragged) tile is element-wise on purpose
]


@pytest.mark.parametrize("fname,needle,max_waiting", CASES)
def test_pipelined_kernels_keep_their_loads_in_flight(asm, fname, needle, max_waiting):
    text, spills = asm[fname]
    ks = kernels(text, needle)
    assert ks, f"{needle}: kernel not found in {fname}.s"
    for name, lines in ks.items():
        waiting, flat = shape(lines)
        assert flat == 0, (name, "flat memory instructions", flat)
        assert spills.get(name, 0) == 0, (name, "VGPRs spilled", spills.get(name))
        assert waiting <= max_waiting, (name, "loads waited for where they are issued", waiting)
