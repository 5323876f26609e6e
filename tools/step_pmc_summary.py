#!/usr/bin/env python3
"""Joins the passes of tools/step_pmc.sh: per kernel, launches, average duration, SQ counters per launch, and the
VALU issue fraction  = SQ_INSTS_VALU (wave instructions) * 2 cycles / (1024 SIMDs * clock * duration)   [clock from
GRBM_GUI_ACTIVE / duration], HBM bytes (FETCH_SIZE KiB x 2 on gfx950, WRITE_SIZE KiB)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

SIMDS = 1024


def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*$', '', name)
    return name.replace('ddspp::', '')


def pmc(path):
    agg = defaultdict(lambda: defaultdict(list))
    for fn in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                agg[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    return agg


def main(out):
    dur, calls = {}, {}
    for fn in glob.glob(os.path.join(out, 'kt', '**', '*kernel_stats.csv'), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                k = short(row['Name'])
                dur[k] = dur.get(k, 0.0) + float(row['TotalDurationNs'])
                calls[k] = calls.get(k, 0) + int(row['Calls'])
    ctr = defaultdict(dict)
    for sub in ('a', 'b', 'c', 'f', 'w'):
        for k, cs in pmc(os.path.join(out, sub)).items():
            for c, v in cs.items():
                ctr[k][c] = sum(v) / len(v)
    for k in sorted(dur, key=lambda k: -dur[k]):
        c = ctr.get(k)
        if not c:
            continue
        ms = dur[k] / calls[k] / 1e6
        print(f'{k[:96]}\n    launches {calls[k]}  avg {ms:.4f} ms')
        gui = c.get('GRBM_GUI_ACTIVE')
        if gui:
            gui /= 8.0            # the counter is summed over the 8 XCDs
        clk = gui / (ms * 1e-3) if gui else None
        for name in sorted(c):
            print(f'    {name:30s} {c[name]:18.1f}')
        if clk and 'SQ_INSTS_VALU' in c:
            # kernel time under the counter pass differs slightly from the trace pass: use the pass's own GRBM cycles
            frac = c['SQ_INSTS_VALU'] * 2.0 / (SIMDS * gui)
            print(f'    -> clock {clk / 1e9:.3f} GHz; VALU issue fraction (2 cycles per wave64 instruction) = {frac:.3f}')
            if c.get('SQ_INSTS_VALU_TRANS_F32'):
                tr = c['SQ_INSTS_VALU_TRANS_F32']
                mix = ((c['SQ_INSTS_VALU'] - tr) * 2.0 + tr * 8.0) / (SIMDS * gui)
                print(f'    -> with the {tr / c["SQ_INSTS_VALU"]:.1%} quarter-rate transcendentals at 8 cycles: {mix:.3f} of the issue cycles')
        if 'SQ_ACTIVE_INST_VALU' in c and 'SQ_BUSY_CYCLES' in c and gui:
            print(f'    -> SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY = {c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_ACTIVE_INST_ANY", 1.0), 1.0):.2f}')
        if 'FETCH_SIZE' in c or 'WRITE_SIZE' in c:
            rd = 2.0 * c.get('FETCH_SIZE', 0.0) * 1024 / 1e9
            wr = c.get('WRITE_SIZE', 0.0) * 1024 / 1e9
            print(f'    -> HBM read {rd:.3f} GB, written {wr:.3f} GB per launch = {(rd + wr) / (ms * 1e-3) / 1e3:.2f} TB/s')


    # wave-instruction count of one ddspp_polyphonic_additive call (bench.py roofline_step reads it from profiles/)
    add_kernels = [k for k in ctr if k.startswith(('osc_prepass', 'osc_count', 'bank_slot_sum', 'osc_offset_scan', 'bank_compact'))]
    tot = sum(ctr[k].get('SQ_INSTS_VALU', 0.0) for k in add_kernels)
    if tot:
        import json
        trans = sum(ctr[k].get('SQ_INSTS_VALU_TRANS_F32', 0.0) for k in add_kernels)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from ddsp_piano_amd import _lib
        # ... and of the FilteredNoise kernel(s) (roofline_noise)
        nz = [k for k in ctr if k.startswith(('noise_win_fused', 'noise_fir_fused', 'tv_fir', 'fir_design'))]
        noise = {name: sum(ctr[k].get(c, 0.0) for k in nz) for name, c in
                 (('valu', 'SQ_INSTS_VALU'), ('fma', 'SQ_INSTS_VALU_FMA_F32'), ('trans', 'SQ_INSTS_VALU_TRANS_F32'),
                  ('mfma_mops', 'SQ_INSTS_VALU_MFMA_MOPS_F32'), ('lds', 'SQ_INSTS_LDS'))}
        noise['kernels'] = nz
        json.dump({'csrc_hash': _lib.source_hash(),
                   'valu_wave_instructions_per_call': tot, 'valu_trans_wave_instructions_per_call': trans,
                   'kernels': {k: ctr[k].get('SQ_INSTS_VALU', 0.0) for k in add_kernels},
                   'noise': noise,
                   'source': 'rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_VALU_TRANS_F32 over bench.py --steps 3 --warmup 1 (tools/step_pmc.sh), '
                             'per-launch averages of the kernels of ddspp_polyphonic_additive at BASELINE config 3'},
                  open(os.path.join(out, 'step_valu.json'), 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1])
