#!/usr/bin/env python3
"""Generates the straight-line asm of lz_links' inserter for one whole block (32 steps = 16 pairs), software-pipelined:
per pair i -- read the bucket addresses of pair i + AHEAD, write the links of pair i - BEHIND, exchange pair i -- with the
counted wait that makes "the exchanges of pair i - BEHIND are back, the addresses of pair i are here" one s_waitcnt.
usage: python tools/gen_links_block.py [AHEAD BEHIND]   (prints the lines to paste between the asm's quotes)"""
import sys
AHEAD = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BEHIND = int(sys.argv[2]) if len(sys.argv) > 2 else 3
PAIRS = 16
NA, NQ = AHEAD + 1, BEHIND + 1
A = [(100 + 2 * k, 101 + 2 * k) for k in range(NA)]                 # address register pairs
Q0 = 100 + 2 * NA
Q = [(Q0 + 2 * k, Q0 + 2 * k + 1) for k in range(NQ)]               # result register pairs
P0, P1 = Q0 + 2 * NQ, Q0 + 2 * NQ + 1                               # positions of the pair's two steps
DUMMY = P1 + 1                                                      # (unused: no dummy operations, the waits are counted per site)
BASE0 = DUMMY                                                   # base + 0x400 j, j = 1..7
assert BASE0 + 6 <= 127
def base(i):
    j = i >> 1
    return "%[base]" if j == 0 else "v%d" % (BASE0 + j - 1)
def offs(i):
    return "offset1:64" if (i & 1) == 0 else "offset0:128 offset1:192"
out = []
issued = []                                       # LDS operations in issue order (they complete in order): names
def emit(s): out.append('                "%s\\n\\t"' % s)
def lds(name, text): issued.append(name); emit(text)
def wait_for(*names):                             # everything up to the newest of `names` has completed
    need = max(len(issued) - 1 - issued[::-1].index(n) for n in names if n in issued)
    emit("s_waitcnt lgkmcnt(%d)" % (len(issued) - 1 - need))
for j in range(1, 8): emit("v_add_u32 v%d, 0x%x, %%[base]" % (BASE0 + j - 1, 0x400 * j))
emit("v_mov_b32 v%d, %%[p0]" % P0)
emit("v_add_u32 v%d, 0x40, %%[p0]" % P1)
def R(i): lds("R%d" % i, "ds_read2_b32 v[%d:%d], %s %s" % (A[i % NA][0], A[i % NA][1], base(i), offs(i)))
def W(i): lds("W%d" % i, "ds_write2_b32 %s, v%d, v%d %s" % (base(i), Q[i % NQ][0], Q[i % NQ][1], offs(i)))
for i in range(min(AHEAD, PAIRS)): R(i)
for i in range(PAIRS):
    if i + AHEAD < PAIRS: R(i + AHEAD)
    wait_for("R%d" % i, "X%d" % (i - BEHIND))     # the addresses of pair i; the exchanges whose links go out now (and whose registers pair i + 1 ... reuse)
    if i - BEHIND >= 0: W(i - BEHIND)
    lds("x%d" % i, "ds_wrxchg_rtn_b32 v%d, v%d, v%d" % (Q[i % NQ][0], A[i % NA][0], P0))
    lds("X%d" % i, "ds_wrxchg_rtn_b32 v%d, v%d, v%d" % (Q[i % NQ][1], A[i % NA][1], P1))
    if i + 1 < PAIRS:
        emit("v_add_u32 v%d, 0x80, v%d" % (P0, P0))
        emit("v_add_u32 v%d, 0x80, v%d" % (P1, P1))
for i in range(max(0, PAIRS - BEHIND), PAIRS):     # drain: exchanged, not written yet
    wait_for("X%d" % i)
    W(i)
out.append('                "s_nop 0"')
print("\n".join(out))
print("// clobbers: " + ", ".join('"v%d"' % r for r in list(range(100, DUMMY)) + list(range(BASE0, BASE0 + 7))), file=sys.stderr)
