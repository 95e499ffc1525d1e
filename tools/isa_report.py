#!/usr/bin/env python3
"""Per-function register / scratch / packed-op summary of a gfx950 ISA listing (hipcc -S --cuda-device-only).

usage: python tools/isa_report.py [file.s]   (default: compiles longcalld_amd/csrc/poa_kernel.hip to /tmp/poa_kernel.s)
Prints, per function of the listing: VGPRs, scratch bytes per lane, scratch_ instructions, v_pk_ instructions, code bytes;
and the totals the round's targets are stated in (grep -c scratch_, .vgpr_spill_count of every kernel)."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), text=True, capture_output=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = "/tmp/poa_kernel.s"
        src = os.path.join(ROOT, "longcalld_amd", "csrc", "poa_kernel.hip")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-w", "-o", path, src])
    funcs, cur = [], None
    for line in open(path):
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            cur = dict(name=m.group(1), scratch_i=0, pk=0, vgpr=None, scratch=None, code=None, insts=0)
            funcs.append(cur)
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith("scratch_"):
            cur["scratch_i"] += 1
        if s.startswith("v_pk_"):
            cur["pk"] += 1
        if s and not s.startswith((";", ".", "//")) and not s.endswith(":"):
            cur["insts"] += 1
        m = re.match(r"; NumVgprs: (\d+)", s)
        if m: cur["vgpr"] = int(m.group(1))
        m = re.match(r"; ScratchSize: (\d+)", s)
        if m: cur["scratch"] = int(m.group(1))
        m = re.match(r"; codeLenInByte = (\d+)", s)
        if m: cur["code"] = int(m.group(1))
    funcs = [f for f in funcs if f["vgpr"] is not None]
    dm = demangle([f["name"] for f in funcs])
    print(f"{'VGPR':>5} {'scratchB':>8} {'scratch_':>8} {'v_pk_':>6} {'codeB':>7}  function")
    for f in funcs:
        n = dm.get(f["name"], f["name"])
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)
        print(f"{f['vgpr']:5d} {f['scratch']:8d} {f['scratch_i']:8d} {f['pk']:6d} {f['code']:7d}  {n}")
    print("total scratch_ instructions:", sum(f["scratch_i"] for f in funcs), " total v_pk_:", sum(f["pk"] for f in funcs))
    txt = open(path).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count: (\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count: (\d+)", txt):
        if "poa_chain" in m.group(1):
            print("kernel", re.sub(r"EEv.*", "", m.group(1)), "private", m.group(2), "sgpr_spill", m.group(3), "vgprs", m.group(4), "vgpr_spill", m.group(5))


if __name__ == "__main__":
    main()
