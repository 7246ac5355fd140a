#!/usr/bin/env python3
"""Per-kernel resource table of the built library: VGPRs, AGPRs, SGPRs, spills, scratch bytes per lane, LDS, workgroup size and the
scratch instructions that sit inside loops (a backward branch spans them) -- read from the gfx950 code object inside
uglad_amd/csrc/libuglad_hip.so with the ROCm LLVM tools; nothing is run on a GPU.

    python scripts/kernel_meta.py [--so PATH] [--filter SUBSTR] [--loops] > profiles/rNN_kernel_meta.txt

--loops disassembles every kernel that has scratch and counts scratch_load / scratch_store instructions inside backward-branch spans.
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def extract_code_object(so: str, out_dir: str) -> str:
    """The gfx950 ELF bundled into the host library."""
    co = os.path.join(out_dir, "gfx950.co")
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={so}"], capture_output=True, text=True)
    targets = [t for t in r.stdout.split() if "gfx950" in t]
    if targets:
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={so}", f"--targets={targets[0]}",
                        f"--output={co}"], check=True, capture_output=True)
        if os.path.getsize(co) > 0:
            return co
    # a linked .so keeps the fat binary in section .hip_fatbin: cut it out and unbundle that
    fat = os.path.join(out_dir, "fatbin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    objs = []
    pos = 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = int.from_bytes(data[i + 24:i + 32], "little")
        p = i + 32
        for _ in range(n):
            off = int.from_bytes(data[p:p + 8], "little")
            size = int.from_bytes(data[p + 8:p + 16], "little")
            tlen = int.from_bytes(data[p + 16:p + 24], "little")
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                objs.append(data[i + off:i + off + size])
        pos = i + 24
    if not objs:
        raise SystemExit(f"no gfx950 code object in {so}")
    paths = []
    for k, o in enumerate(objs):
        pk = os.path.join(out_dir, f"gfx950_{k}.co")
        open(pk, "wb").write(o)
        paths.append(pk)
    return paths


def notes(co: str):
    """Kernel descriptors' metadata (msgpack rendered as YAML by llvm-readelf --notes)."""
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    kernels = []
    cur = None
    for line in txt.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "agpr_count" or (key == "args" and cur is None):
            pass
        if line.lstrip().startswith("- .") and key in ("agpr_count", "args"):
            cur = {}
            kernels.append(cur)
        if cur is not None and key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                       "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size",
                                       "name", "uses_dynamic_stack"):
            cur[key] = val.strip("'\"")
    return [k for k in kernels if "name" in k]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.splitlines()
    return dict(zip(names, out)) if len(out) == len(names) else {n: n for n in names}


def scratch_in_loops(co: str, sym: str):
    """(scratch instructions, of them inside a backward-branch span, instructions) for one kernel symbol."""
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={sym}", co], capture_output=True, text=True).stdout
    ins = []  # (addr, mnemonic, target or None)
    for line in txt.splitlines():
        m = re.match(r"\s+(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        addr = int(m.group(3), 16)
        mn = m.group(1)
        tgt = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            mt = re.search(r"<[^>+]+\+0x([0-9a-fA-F]+)>", line)
            if mt:
                tgt = int(mt.group(1), 16)
        ins.append((addr, mn, tgt))
    if not ins:
        return 0, 0, 0
    base = ins[0][0]
    spans = [(base + t, a) for a, mn, t in ins if t is not None and base + t <= a]
    scr = [a for a, mn, _ in ins if mn.startswith("scratch_")]
    inside = sum(1 for a in scr if any(lo <= a <= hi for lo, hi in spans))
    return len(scr), inside, len(ins)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "uglad_amd", "csrc", "libuglad_hip.so"))
    ap.add_argument("--filter", default="")
    ap.add_argument("--loops", action="store_true")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        cos = extract_code_object(a.so, td)
        if isinstance(cos, str):
            cos = [cos]
        rows = []
        for co in cos:
            ks = notes(co)
            dm = demangle([k["name"] for k in ks])
            for k in ks:
                k["pretty"] = re.sub(r"\(.*$", "", dm[k["name"]]).replace("void ", "")
                k["co"] = co
                rows.append(k)
        rows = [k for k in rows if a.filter in k["pretty"]]
        rows.sort(key=lambda k: k["pretty"])
        print(f"# {os.path.relpath(a.so, ROOT)}: {len(rows)} kernels in {len(cos)} gfx950 code object(s)")
        hdr = f"{'kernel':58s} {'wg':>5s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s} {'ldsB':>7s}"
        if a.loops:
            hdr += f" {'scr_ins':>7s} {'in_loops':>8s}"
        print(hdr)
        for k in rows:
            line = (f"{k['pretty'][:58]:58s} {k.get('max_flat_workgroup_size', '?'):>5s} {k.get('vgpr_count', '?'):>5s} "
                    f"{k.get('agpr_count', '?'):>5s} {k.get('sgpr_count', '?'):>5s} {k.get('vgpr_spill_count', '0'):>6s} "
                    f"{k.get('sgpr_spill_count', '0'):>6s} {k.get('private_segment_fixed_size', '0'):>8s} "
                    f"{k.get('group_segment_fixed_size', '0'):>7s}")
            if a.loops:
                if int(k.get("private_segment_fixed_size", "0") or 0) > 0:
                    n, inside, tot = scratch_in_loops(k["co"], k["name"])
                    line += f" {n:7d} {inside:8d}"
                else:
                    line += f" {0:7d} {0:8d}"
            print(line)


if __name__ == "__main__":
    sys.exit(main())
