#!/usr/bin/env python3
"""Regenerate the round-4 number paragraph of DESIGN.md section 7 and the result tables of BASELINE.md section 8 from
profiles/r04_bench.json (+ r04_bench_under_rocprof.json, r04_bench_kernel_stats.csv, osc_traffic.json), so that the prose
quotes the committed bench line and nothing else.  usage: python tools/results_tables.py"""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return json.loads(open(os.path.join(ROOT, 'profiles', name)).read().strip().splitlines()[-1])


d = load('r04_bench.json')
under = load('r04_bench_under_rocprof.json')
traffic = json.load(open(os.path.join(ROOT, 'profiles', 'osc_traffic.json')))
trace_ms = None
for r in csv.DictReader(open(os.path.join(ROOT, 'profiles', 'r04_bench_kernel_stats.csv'))):
    if r['Name'].startswith('void ddspp::osc_kernel<1, false, 0, true'):
        trace_ms = float(r['AverageNs']) / 1e6
ms = lambda k: d[k]['ms_per_step']['median']          # noqa: E731
seg = lambda k: d[k]['ms_per_segment']                # noqa: E731
sc = d['shipped_configs']
ro, rs, rn = d['roofline'], d['roofline_step'], d['roofline_noise']
sr = 24000.0
B, N = 64, 72000

design = (
    f"**{d['ms_per_step']:.3f} ms per step** = {d['value'] / 1e9:.2f}e9 samples/s = {d['rtf'] / 1e3:.1f}k × real time (`pipelined` "
    f"{d['pipelined']['ms_per_step']:.2f} ms; earlier trees of the round on other boxes:\n"
    "1.624, 1.626, 1.629, 1.657; mid-round, before the lean get_controls kernel, the trimmed walk, the reverb's prefetches, the hoisted scale_fn\n"
    "dispatch and the bank's per-path block loops: 1.850);\n"
    f"audio only {ms('audio_only_call'):.3f} ms; the `--decompose` sums (`group.decompose`) {ms('decompose_call'):.3f} ms; every voice's stems "
    f"(`need_stems=True`: per-voice rows, nothing compacted) {ms('all_stems_call'):.2f} ms; every f0 moving\n"
    f"**{ms('moving_f0'):.2f} ms**, dense worst case **{ms('dense_worst_case'):.2f} ms** (3.34 / 5.78 mid-round: the bank's per-path loops and the "
    "frame-wise pre-pass scan);\n"
    f"one 3 s segment {seg('single_stream'):.3f} ms (Python), {seg('single_stream_native'):.3f} (one-call driver), "
    f"{seg('single_stream_graph_native'):.3f} (its kernels as a replayed graph, controls written in\n"
    f"place), {seg('single_stream_graph'):.3f} (graph over the Python group with input copies); 136 s file "
    f"{d['whole_file']['ms_per_file']:.2f} ms;\n"
    f"**20-minute file (300 000 frames) {d['whole_file_20min']['ms_per_file']:.1f} ms = {d['whole_file_20min']['rtf'] / 1e3:.0f}k × real time** "
    "(before round 4 it left the fused kernels at frame 131 072);\n"
    f"config 5's per-GPU share **{ms('c5_per_gpu_share'):.2f} ms** (round 3: 3.46–3.58); dafx22 dims {ms('dafx22_dims'):.3f} ms and "
    f"**default_model.py's node list {ms('default_model_dag'):.3f} ms** (it\n"
    "used to walk 4 P nodes one eager call at a time); **every shipped gin file at its own dims and flags** (`shipped_configs`, batch 64\n"
    "× 3 s, poly 16): " + ", ".join(f"{k} {'**' if k == 'surrogate' else ''}{v['ms_per_step']['median']:.2f}{'**' if k == 'surrogate' else ''}"
                                   for k, v in sc.items()) + " ms\n"
    "(surrogate: 19.2 ms at batch 16, i.e. about 77 at this batch, before SurrogateAdditive reached the fused kernels, §14);\n"
    f"graded kernel {ro['ms_per_launch']:.2f} ms = {ro['achieved'] / 1e3:.2f} TB/s = **{ro['frac']:.3f} of 8 TB/s** = "
    f"{ro['frac_of_measured_peak']:.3f} of the pure\n"
    "read of the same buffers (measured first in the process since the last commits: 0.726–0.737 on a fresh heap against\n"
    "0.717–0.726 behind the bench's other sections, `tools/roofline_order.py`; §4a: 0.70–0.77 from process to process), PMC traffic /\n"
    f"algorithmic = {traffic['traffic_over_algorithmic']:.6f}; FilteredNoise call {rn['ms_per_call']:.3f} ms, `ddspp_polyphonic_additive` "
    f"{rs['ms_per_call']:.3f} (`frac_mix` {rs.get('frac_mix', float('nan')):.2f}); numpy oracle on the\n"
    f"host {d['cpu_baseline']['value'] / 1e4:.1f}e4 samples/s ({d['cpu_baseline']['cores']} threads).  `profiles/r04_bench_kernel_stats.csv` / "
    "`r04_step_pmc.txt` / `step_valu.json` /\n"
    "`osc_traffic.json` are the counter passes of the same tree and the same call of `tools/profile_r04.sh` (source hash\n"
    f"`{traffic['csrc_hash']}`; the bench line inside the kernel trace: graded kernel {under['roofline']['ms_per_launch']:.2f} ms by HIP "
    f"events, {trace_ms:.2f} ms in the trace; this paragraph and BASELINE.md §8's tables: `tools/results_tables.py`).\n")

p = os.path.join(ROOT, 'DESIGN.md')
s = open(p).read()
a = s.index('**', s.index('Round-4 numbers (MI355X, `profiles/r04_bench.json`'))
b = s.index("Same-box A/B of the round's kernel changes: §4a")
open(p, 'w').write(s[:a] + design + s[b:])


def row(label, form, t_ms, samples, note, bold=False):
    v = samples / (t_ms * 1e-3)
    t = f'**{t_ms:.3g}**' if bold else f'{t_ms:.4g}'
    return f'| {label} | {form} | {t} | {v / 1e9:.2f}e9 | {v / sr / 1e3:.1f}k x | {note} |'


c5, dx = d['c5_per_gpu_share'], d['dafx22_dims']
rows = [
    row('C3 B=64 (SURVEY §8d inputs)', 'outputs dict (headline, synchronised median)', d['ms_per_step'], B * N,
        f"`pipelined`: {d['pipelined']['ms_per_step']:.2f} ms; earlier trees of the round on other boxes 1.624, 1.626, 1.629, 1.657; mid-round 1.850; "
        "round 3 on its (fast) box: 1.825", True),
    row('C3 B=64', 'audio only', ms('audio_only_call'), B * N, ''),
    row('C3 B=64', '`group.decompose(features)`: output, dry mix, the two sums over the voices', ms('decompose_call'), B * N,
        'new: `synthesize_from_csv.py:92-120` in one batched call'),
    row("C3 B=64, all 16 voices' stems", 'outputs dict, `need_stems=True`', ms('all_stems_call'), B * N,
        'per-voice rows through the fused kernels, nothing compacted'),
    row('C3 B=64, every f0 moving', 'outputs dict', ms('moving_f0'), B * N,
        "mid-round 3.34: the bank's per-path block loops, the frame-wise pre-pass scan", True),
    row('C3 B=64, dense worst case', 'outputs dict', ms('dense_worst_case'), B * N, 'mid-round 5.78', True),
    row('C2 one 3 s poly-16 segment', 'outputs dict', seg('single_stream'), N,
        f"{seg('single_stream_native'):.3f} one-call driver, {seg('single_stream_graph_native'):.3f} its kernels as a replayed graph with "
        "controls in place"),
    row('whole file 136 s, poly 16, B=1', 'outputs dict', d['whole_file']['ms_per_file'], 34000 * 96, ''),
    row('**whole file 20 min (300 000 frames), poly 16, B=1**', 'outputs dict', d['whole_file_20min']['ms_per_file'], 300000 * 96,
        'new: past 131 072 frames a file used to leave the fused kernels (DESIGN.md §13)', True),
]
rows.append(f"| C5 per-GPU share: 48 kHz, poly 32, H=128, K=96, 10 s IR, B=32 | outputs dict | **{ms('c5_per_gpu_share'):.3g}** | "
            f"{c5['value'] / 1e9:.2f}e9 | {c5['rtf'] / 1e3:.1f}k x | the windowed FilteredNoise kernel has a hop-192 instance now, then the lean "
            "get_controls kernel (round 3: 3.46–3.58) |")
rows.append(f"| dafx22 dims: 16 kHz, poly 16, H=96, K=64, S=2, 1.5 s IR, B=64 | outputs dict | {ms('dafx22_dims'):.4g} | {dx['value'] / 1e9:.2f}e9 | "
            f"{dx['rtf'] / 1e3:.1f}k x | |")
rows.append(f"| **the same through `default_model.py:44-80`'s node list** (explicit Add nodes) | outputs dict | **{ms('default_model_dag'):.4g}** | "
            f"{64 * 48000 / (ms('default_model_dag') * 1e-3) / 1e9:.2f}e9 | {d['default_model_dag']['rtf'] / 1e3:.1f}k x | new: batched route (it "
            "walked 4 P nodes eagerly) |")
rows.append(f"| graded kernel `ddspp_cos_oscillator_bank`, 75.8 GB | – | {ro['ms_per_launch']:.2f} | – | – | {ro['achieved'] / 1e3:.2f} TB/s = "
            f"**{ro['frac']:.3f} of 8 TB/s** = {ro['frac_of_measured_peak']:.3f} of the pure read of the same buffers ({ro['measured_peak'] / 1e3:.2f} "
            f"TB/s, `ddspp_hbm_read_probe`) = {ro['frac_of_guide_achievable']:.3f} of the guide's 6.29 TB/s; PMC traffic / algorithmic = "
            f"{traffic['traffic_over_algorithmic']:.6f} |")
rows.append(f"| FilteredNoise call / `ddspp_polyphonic_additive` / get_controls alone | – | {rn['ms_per_call']:.3f} / {rs['ms_per_call']:.3f} / 0.118 | – | – | "
            "socket power while looping: 1391–1402 W at 2.19–2.27 GHz / 1120–1130 W at 2.39 GHz |")
cb = d['cpu_baseline']
rows.append(f"| CPU: numpy oracle, {cb['cores']} threads / torch-CPU chain | – | – | {cb['value'] / 1e4:.1f}e4 / "
            f"{cb['torch_cpu_all_cores']['value'] / 1e4:.1f}e4 | {cb['rtf']:.1f} x | \"restatement baseline — TF/ddsp unavailable\" |")
shipped = []
for k, v in sc.items():
    w = v['workload'].split(' (')[0]
    srk = int(re.search(r'(\d+) Hz', w).group(1))
    n = 64 * 750 * (srk // 250)
    shipped.append(f"| `{k}.gin` | {w} | **{v['ms_per_step']['median']:.2f}** | {n / (v['ms_per_step']['median'] * 1e-3) / 1e9:.2f}e9 | "
                   f"{v['rtf'] / 1e3:.1f}k x |")
p = os.path.join(ROOT, 'BASELINE.md')
s = open(p).read()
h = s.index('| config / input | call form | ms per step | samples/s | RTF | note |', s.index('## 8. Results (round 4'))
e = s.index('`surrogate.gin` walked its voices one by one', h)
block = ('| config / input | call form | ms per step | samples/s | RTF | note |\n|---|---|---|---|---|---|\n' + '\n'.join(rows) +
         '\n\nEvery shipped gin file at its own dims and flags (`shipped_configs` in the bench line; `dafx22.gin` is the dafx22 row above):\n\n'
         '| config | workload | ms per step | samples/s | RTF |\n|---|---|---|---|---|\n' + '\n'.join(shipped) + '\n\n')
open(p, 'w').write(s[:h] + block + s[e:])
print('DESIGN.md section 7 and BASELINE.md section 8 regenerated from profiles/r04_bench.json')
