#!/usr/bin/env python3
"""One line per kernel of a hipcc -S listing: instructions, VALU, moves (v_mov_b32 / v_mov_b64: register shuffling),
branches (scalar + EXEC), `s_waitcnt vmcnt(0)` next to global loads (loads serialised inside branches), scratch
traffic, registers, spills.  What found get_controls' fat in round 4.
usage: python tools/isa_audit.py file.s [name filter]"""
import collections
import re
import sys

src = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
meta = {}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', src, re.S):
    g = lambda k: int(re.search(r'\.' + k + r':\s+(\d+)', m.group(2)).group(1))  # noqa: E731
    meta[m.group(1)] = (g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'))
print(f"{'kernel':72s} {'instr':>6s} {'valu':>6s} {'mov32':>5s} {'mov64':>5s} {'sbr':>4s} {'execz':>5s} {'vm0':>4s} {'gld':>4s} {'gst':>4s} {'scr':>4s} {'vgpr':>4s} {'spill':>5s} {'sspill':>6s}")
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', src, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if name not in meta or flt not in name:
        continue
    ops = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in ';.' or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    tot = sum(ops.values())
    valu = sum(v for k, v in ops.items() if k.startswith('v_'))
    sbr = sum(v for k, v in ops.items() if k.startswith('s_cbranch_scc') or k.startswith('s_cbranch_vcc') or k == 's_branch')
    execz = ops['s_cbranch_execz'] + ops['s_cbranch_execnz']
    vm0 = len(re.findall(r's_waitcnt vmcnt\(0\)', body))
    gld = sum(v for k, v in ops.items() if k.startswith('global_load') or k.startswith('buffer_load') or k.startswith('flat_load'))
    gst = sum(v for k, v in ops.items() if k.startswith('global_store') or k.startswith('buffer_store') or k.startswith('flat_store'))
    scr = sum(v for k, v in ops.items() if k.startswith('scratch_'))
    v, sp, ssp = meta[name]
    short = re.sub(r'^_ZN5ddspp\d*', '', name)[:72]
    print(f'{short:72s} {tot:6d} {valu:6d} {ops["v_mov_b32_e32"]:5d} {ops["v_mov_b64_e32"]:5d} {sbr:4d} {execz:5d} {vm0:4d} {gld:4d} {gst:4d} {scr:4d} {v:4d} {sp:5d} {ssp:6d}')
