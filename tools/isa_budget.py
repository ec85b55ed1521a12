#!/usr/bin/env python3
"""Instruction budget of a kernel from the compiler's own listing (no GPU needed).

    isa_budget.py extract <mangled-name-substring> [-D...]      compile csrc/aqua_hip.hip device-only with the product's
                                                               flags, write the kernel's listing to stdout
    isa_budget.py blocks  <kernel.s>                            one line per basic block: index, label, counts, first comment
    isa_budget.py count   <kernel.s> name=i,j,k-m ...           per named path (lists / ranges of block indices): instructions
                                                               by class and vector-issue cycles
    isa_budget.py layout  <kernel.s> name=i,j,k-m ...           where the code lies: byte offset and size of every basic block
                                                               (llvm-mc's encodings of the listing), and per named path the
                                                               bytes it executes, the span they lie in and the 64-byte
                                                               instruction-cache lines it touches

Vector-issue cycles of a wavefront on its SIMD: 4 per VALU instruction (64 lanes on 16), 16 for the quarter-rate ones
(32-bit integer multiplies incl. v_mad_u64_u32, transcendentals v_sin/cos/sqrt/rcp/rsq/exp/log, 64-bit float divides /
square roots), 8 for the other float64 arithmetic and 64-bit shifts (MI355X_MICROARCH.md, CDNA4 ISA guide).  Scalar,
memory and LDS instructions issue on other ports and are listed beside them.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUARTER = re.compile(r"^v_(mad_u64_u32|mad_i64_i32|mul_lo_u32|mul_hi_u32|mul_hi_i32|mul_lo_i32|sin_f32|cos_f32|sqrt_f32|rcp_f32|"
                     r"rsq_f32|exp_f32|log_f32|rcp_f64|rsq_f64|sqrt_f64|div_fixup_f64|div_fmas_f64|div_scale_f64)")
HALF = re.compile(r"^v_(\w+_f64|lshlrev_b64|lshrrev_b64|ashrrev_i64|mul_f64|fma_f64|add_f64)")


def klass(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm" or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "branch"
    if op.startswith("s_waitcnt") or op == "s_nop" or op.startswith("s_barrier") or op.startswith("s_sleep") or op.startswith("s_setprio"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_load") or op.startswith("flat_load") or op.startswith("buffer_load") or op.startswith("scratch_load"):
        return "vmem_rd"
    if op.startswith("global_store") or op.startswith("flat_store") or op.startswith("buffer_store") or op.startswith("scratch_store") \
            or op.startswith("global_atomic") or op.startswith("flat_atomic"):
        return "vmem_wr"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def cycles(op):
    if klass(op) not in ("valu", "lane"):
        return 0
    if QUARTER.match(op):
        return 16
    if HALF.match(op):
        return 8
    return 4


def parse(path):
    """-> list of blocks: dict(label, comment, ops=[mnemonic, ...])"""
    blocks = [dict(label="entry", comment="", ops=[])]
    for raw in open(path):
        line = raw.rstrip("\n")
        text = line.strip()
        if not text or text.startswith(";;#") or text.startswith(".") and not text.startswith(".LBB"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", text)
        if m:
            blocks.append(dict(label=m.group(1), comment=(m.group(2) or "").strip("; "), ops=[]))
            continue
        m = re.match(r"^; %bb\.(\d+):\s*(;.*)?$", text)
        if m:
            blocks.append(dict(label="bb.%s" % m.group(1), comment=(m.group(2) or "").strip("; "), ops=[]))
            continue
        if text.startswith(";") or text.endswith(":"):
            continue
        op = text.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            blocks[-1]["ops"].append(op)
    return blocks


def tally(ops):
    t = {}
    for op in ops:
        k = klass(op)
        t[k] = t.get(k, 0) + 1
    t["valu_issue_cycles"] = sum(cycles(op) for op in ops)
    t["quarter_rate"] = sum(1 for op in ops if QUARTER.match(op))
    return t


def indices(spec, n):
    out = []
    for part in spec.split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    assert all(0 <= i < n for i in out), "block index out of range"
    return out


def block_bytes(path):
    """-> bytes of code per basic block, same block numbering as parse(): the listing through llvm-mc -show-encoding."""
    mc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-mc")
    out = subprocess.run([mc, "--triple=amdgcn-amd-amdhsa", "--mcpu=gfx950", "-show-encoding", path],
                         capture_output=True, text=True, check=True).stdout
    # llvm-mc drops the "; %bb.N:" comments: take the block boundaries from parse() by instruction count instead
    sizes = []
    for line in out.splitlines():
        m = re.search(r"; encoding: \[([^\]]*)\]", line)
        if m and re.match(r"^\s+[a-z_0-9]+(\s|$)", line):
            sizes.append(len(m.group(1).split(",")))
    blocks = parse(path)
    n_ops = sum(len(b["ops"]) for b in blocks)
    assert n_ops == len(sizes), "llvm-mc saw %d instructions, the listing holds %d" % (len(sizes), n_ops)
    per_block, at = [], 0
    for b in blocks:
        per_block.append(sum(sizes[at:at + len(b["ops"])]))
        at += len(b["ops"])
    return blocks, per_block


def layout(path, specs):
    blocks, size = block_bytes(path)
    start = [0]
    for s in size:
        start.append(start[-1] + s)
    print("kernel: %d basic blocks, %d bytes of code = %d instruction-cache lines of 64 B" % (len(blocks), start[-1], (start[-1] + 63) // 64))
    for spec in specs:
        name, idx = spec.split("=")
        ids = sorted(set(indices(idx, len(blocks))))
        lines = set()
        for i in ids:
            if size[i]:
                lines.update(range(start[i] // 64, (start[i] + size[i] - 1) // 64 + 1))
        executed = sum(size[i] for i in ids)
        lo, hi = min(start[i] for i in ids), max(start[i] + size[i] for i in ids)
        print("%-36s executes %6d B in [%6d, %6d) = %5.1f %% of that span; %4d cache lines touched (%d B); "
              "skipped inside the span: %d B" % (name, executed, lo, hi, 100.0 * executed / max(hi - lo, 1), len(lines), 64 * len(lines),
                                                  hi - lo - executed))
    print()
    for i, b in enumerate(blocks):
        print("%3d %-10s at %6d  %5d B" % (i, b["label"], start[i], size[i]))


def main(argv):
    if len(argv) < 2:
        sys.exit(__doc__)
    if argv[0] == "extract":
        sys.path.insert(0, ROOT)
        from aquaticgymenv_amd import build
        flags = [f for f in build.COMMON_FLAGS if f != "-shared"] + argv[2:]
        tmp = "/tmp/aqua_isa_budget.s"
        subprocess.check_call([build.hipcc_path(), *flags, "--cuda-device-only", "-S", build.SRC[0], "-o", tmp],
                              stderr=subprocess.DEVNULL)
        on = False
        for line in open(tmp):
            if not on and re.match(r"^_Z\w*:", line) and argv[1] in line.split(":")[0]:
                on = True
            if on:
                sys.stdout.write(line)
                if ".end_amdhsa_kernel" in line:
                    break
        return
    if argv[0] == "layout":
        layout(argv[1], argv[2:])
        return
    blocks = parse(argv[1])
    if argv[0] == "blocks":
        for i, b in enumerate(blocks):
            t = tally(b["ops"])
            print("%3d %-10s valu %3d (q %2d) lane %2d salu %3d smem %2d rd %2d wr %2d lds %2d  %s%s" % (
                i, b["label"], t.get("valu", 0), t["quarter_rate"], t.get("lane", 0), t.get("salu", 0), t.get("smem", 0),
                t.get("vmem_rd", 0), t.get("vmem_wr", 0), t.get("lds", 0), b["comment"][:70],
                "  [%s]" % " ".join(b["ops"][:4]) if not b["comment"] else ""))
        return
    if argv[0] == "count":
        out = {}
        for spec in argv[2:]:
            name, idx = spec.split("=")
            ops = [op for i in indices(idx, len(blocks)) for op in blocks[i]["ops"]]
            out[name] = dict(tally(ops), blocks=idx)
        whole = [op for b in blocks for op in b["ops"]]
        out["_whole_kernel_static"] = tally(whole)
        json.dump(out, sys.stdout, indent=1, sort_keys=True)
        sys.stdout.write("\n")
        return
    sys.exit(__doc__)


if __name__ == "__main__":
    main(sys.argv[1:])
