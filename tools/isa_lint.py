#!/usr/bin/env python3
"""ISA lint for libhsefr.so: refuse a build that contains a store-data hazard hipcc does not guard on gfx950.

Measured on MI355X (tools/store_hazard_probe.hip -> profiles/r01_store_hazard_probe.txt, DESIGN.md lesson 14): a 16-byte
store whose data VGPRs are overwritten by a vector instruction too close behind it sends the NEW value to memory in
some of its dwords (24 % of the stores in a busy kernel).  Wait states needed between the store and the write:

    global_store_dwordx4 / buffer_store_dwordx4 with an immediate soffset : 2   (hipcc's hazard recogniser gives 1)
    buffer_store_dwordx4 with an SGPR soffset                             : 1   (hipcc gives 0: createsVALUHazard
                                                                                 exempts MUBUF stores with a register soffset)

so whether a kernel is correct depends on what the scheduler happens to put behind each store (packed-math writes,
`v_pk_*`, are the ones that bite at the larger distance).

The lint disassembles every gfx950 code object embedded in the library and reports each wide store (buffer/global/flat/
scratch, more than 64 bits of data) followed within TWO wait states by a vector write into its data registers.
Exit status 1 if any is found.  `hse_facerec_tf_amd/csrc/build.sh` runs it after linking, tests/test_abi_cpu.py again.

Second rule (round 3, tools/ashr_pk_probe.hip -> profiles/r03_ashr_pk_probe.txt, DESIGN.md lesson 36): `v_ashr_pk_u8_i32` /
`v_ashr_pk_i8_i32` (new on gfx950) write the LOW 16 bits of their destination and keep the upper half, while hipcc (ROCm 7.2)
selects them for  sat_u8(a >> n) | sat_u8(b >> n) << 8  and then ORs further bytes into bits 16..31 as if they were zero:
the C expression  sat(a>>22) | sat(b>>22)<<8 | sat(c>>22)<<16  returns garbage in its upper half whenever the destination
register held something before.  Any occurrence of these instructions fails the build (clamp through an opaque
`v_med3_i32`, as preprocess.hip's u8_round does).

usage: tools/isa_lint.py [path/to/libhsefr.so]
"""
import os
import re
import subprocess
import struct
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
WAIT_STATES = 2      # instructions needed between a wide store and a vector write of its data (tools/store_hazard_probe.hip)

STORE_RE = re.compile(r"^\s*((?:buffer|global|flat|scratch)_store_dwordx[34])\s+(.*)$")
VREG_RE = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)(?!\d))")


def code_objects(lib):
    """Yield (index, bytes) for every gfx950 device ELF in the library's .hip_fatbin section."""
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f.name])
        blob = open(f.name, "rb").read()
    pos, idx = 0, 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield idx, blob[pos + off:pos + off + size]
                idx += 1
        pos += len(MAGIC)


def regs(tok):
    """Set of ('v'|'a', n) registers named by an operand token such as v[4:7], v12 or a[0:3]."""
    out = set()
    for m in VREG_RE.finditer(tok):
        if m.group(2) is not None:
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(1), int(m.group(4))))
    return out


def store_data_regs(mnemonic, operands):
    ops = [o.strip() for o in operands.split(",")]
    # buffer_store: vdata first; global/flat/scratch_store: address first, data second
    return regs(ops[0] if mnemonic.startswith("buffer_") else ops[1])


LOAD_RE = re.compile(r"^\s*((?:buffer|global|flat|scratch)_load_\w+|ds_read\w*|ds_load\w*|ds_bpermute\w*|ds_permute\w*)\s+(.*)$")


def vector_dest_regs(line):
    """Registers written by a vector ALU instruction (first operand of v_* except compares and stores)."""
    m = re.match(r"^\s*(v_\w+)\s+(.*)$", line)
    if not m or m.group(1).startswith(("v_cmp", "v_cmpx", "v_nop")):
        return set()
    return regs(m.group(2).split(",")[0])


def load_dest_regs(line):
    """Destination of a VMEM / DS load.  NOT a hazard: a load's data returns tens to hundreds of cycles after issue, long
    after the store has read its data registers; the compiler re-uses a store's registers for the next load all the
    time (282 places in the round-2 library, whose full-size every-element tests are bit-exact).  Counted and reported
    so that the claim stays visible; it does not fail the build."""
    m = LOAD_RE.match(line)
    if m and " lds" not in line:                  # LDS-DMA loads have no VGPR destination
        return regs(m.group(2).split(",")[0])
    return set()


HALF_WRITE_RE = re.compile(r"^\s*v_ashr_pk_[iu]8_i32\b")
BRANCH_RE = re.compile(r"^\s*s_c?branch\w*\s+(\S+)")


def scan_listing(dis, counts=None):
    """Findings in one llvm-objdump listing: (symbol, store, writer, registers hit, wait states in between).
    The scan is linear (fall-through), and at every branch the pending window is ALSO carried to the first
    instructions of the branch target (a wide store followed by a taken branch is checked against what it lands on)."""
    # pass 1: instruction lines per symbol with their addresses (llvm-objdump prints "// ADDRESS: raw words" behind every
    # instruction and numeric branch offsets), and the instruction index every label / address points at
    symbols = []          # [name, [lines], {label or address: index}]
    for raw in dis.splitlines():
        lab = re.match(r"^[0-9a-f]+ <([^>]+)>:", raw)
        if lab:
            if not lab.group(1).startswith("L") or not symbols:
                symbols.append([lab.group(1), [], {}])
                if counts is not None:
                    counts[0] += 1
            symbols[-1][2][lab.group(1)] = len(symbols[-1][1])
            continue
        line = raw.split("//")[0].strip()
        if line and symbols:
            am = re.search(r"//\s*([0-9A-Fa-f]{6,}):", raw)
            if am:
                symbols[-1][2][int(am.group(1), 16)] = len(symbols[-1][1])
            symbols[-1][1].append(line)
    addr_of = [{i: a for a, i in labels.items() if isinstance(a, int)} for _, _, labels in symbols]

    def advance(pending, line, kernel, findings):
        dest = vector_dest_regs(line)
        ldest = load_dest_regs(line)
        for st in pending:
            hit = st[1] & dest
            if hit:
                findings.append((kernel, st[0], line, sorted(hit), WAIT_STATES - st[2]))
            if counts is not None and len(counts) > 2 and st[1] & ldest:
                counts[2] += 1
        nop = re.match(r"^s_nop\s+(\d+)", line)
        used = int(nop.group(1)) + 1 if nop else 1
        return [[t, r, w - used] for t, r, w in pending if w - used > 0]

    findings = []
    for si, (kernel, lines, labels) in enumerate(symbols):
        pending = []          # [store text, data registers, wait states still needed]
        for i, line in enumerate(lines):
            if HALF_WRITE_RE.match(line):
                findings.append((kernel, line, "(writes 16 bits of its destination; hipcc assumes the upper half is zero)", [], 0))
            pending = advance(pending, line, kernel, findings)
            m = STORE_RE.match(line)
            if m:
                if counts is not None:
                    counts[1] += 1
                pending.append([line, store_data_regs(m.group(1), m.group(2)), WAIT_STATES])
            b = BRANCH_RE.match(line)
            if b and pending:
                tgt = re.sub(r"^<|>$", "", b.group(1))
                j = labels.get(tgt)
                if j is None and re.fullmatch(r"\d+", tgt) and i in addr_of[si]:
                    off = int(tgt)                      # simm16, in dwords, relative to the next instruction
                    off = off - 65536 if off >= 32768 else off
                    j = labels.get(addr_of[si][i] + 4 + 4 * off)
                if j is None:
                    # unknown target (should not happen inside one function): flag conservatively
                    findings.append((kernel, pending[0][0], line + "   <- branch to an unresolved label with a store window open",
                                     sorted(pending[0][1]), WAIT_STATES - pending[0][2]))
                    continue
                carried = [list(p) for p in pending]
                while carried and j < len(lines):
                    carried = advance(carried, lines[j], kernel, findings)
                    j += 1
    return findings


def lint(lib):
    findings, counts = [], [0, 0, 0]
    for idx, elf in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(elf)
            f.flush()
            dis = subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", f.name], text=True)
        findings += scan_listing(dis, counts)
    lint.load_reuse = counts[2]
    return findings, counts[0], counts[1]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                             "hse_facerec_tf_amd", "libhsefr.so")
    findings, nk, ns = lint(lib)
    for kernel, store, nxt, hit, gap in findings:
        print(f"HAZARD in {kernel}:\n    {store}\n    ... {gap} wait state(s) ...\n    {nxt}\n    overwrites {''.join(f'{c}{i} ' for c, i in hit)}",
              file=sys.stderr)
    print(f"isa_lint: {nk} symbols, {ns} wide stores, {len(findings)} store-data hazards "
          f"({getattr(lint, 'load_reuse', 0)} loads re-use a store's data registers inside the window: benign, see load_dest_regs)")
    if nk == 0 or ns == 0:
        print("isa_lint: nothing was scanned (no gfx950 code object / no wide store found): refusing to pass vacuously", file=sys.stderr)
        return 2
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
