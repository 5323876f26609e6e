"""ddsp_piano/modules/polyphonic_dag.py:5-42 -- node list of the polyphonic ProcessorGroup."""
from __future__ import annotations

from .synths import MultiAdd


def polyphonic_dag(additive, noise, reverb=None,
                   additive_controls=('amps', 'harmonic_distribution', 'f0_hz'),
                   noise_controls=('noise_magnitudes',), reverb_controls=(), n_synths=16):
    add = MultiAdd(name='add')
    additive_controls, noise_controls = list(additive_controls), list(noise_controls)
    dag = [(additive, [c + '_0' for c in additive_controls]),
           (noise, [c + '_0' for c in noise_controls]),
           (add, [noise.name + '/signal', additive.name + '/signal'])]
    for i in range(1, n_synths):
        dag.append((additive, [c + f'_{i}' for c in additive_controls]))
        dag.append((noise, [c + f'_{i}' for c in noise_controls]))
        dag.append((add, ['add/signal', noise.name + '/signal', additive.name + '/signal']))
    if reverb is not None:
        dag.append((reverb, ['add/signal'] + list(reverb_controls)))
    return dag
