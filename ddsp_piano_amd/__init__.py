"""ddsp_piano_amd -- MI355X (gfx950) native synthesis hot path of DDSP-Piano.

Drop-in for the processor group of lrenault/ddsp-piano (ddsp_piano/default_model.py:20-85,
ddsp_piano/modules/polyphonic_dag.py, ddsp_piano/modules/piano_model.py:160): same Processor /
ProcessorGroup call signatures, arithmetic in hand-written HIP kernels behind the C-ABI of
include/ddspp.h; plus the glue on its input edge (piano roll -> conditioning, Parallelizer un-merge).
Nothing else of the reference (control networks, MIDI file / audio I/O, training) is rebuilt here.
"""
from . import core  # noqa: F401
from .core import exp_sigmoid, exp_tanh  # noqa: F401
from .graph import CapturedGroup  # noqa: F401
from .native_group import NativeGroup  # noqa: F401
from .effects import (FeedbackDelayNetwork, FeedbackDelayNetworkApply, MultiInstrumentFeedbackDelayReverb,  # noqa: F401
                      MultiInstrumentReverb, Reverb, fdn_impulse_response)
from .noise_band_net import FilterBank, NoiseBandNetSynth  # noqa: F401
from .midi_encoders import MIDIRoll2Conditioning, ensure_sequence_length, roll_to_conditioning  # noqa: F401
from .parallelizer import Parallelizer  # noqa: F401
from .polyphonic_dag import polyphonic_dag  # noqa: F401
from .processors import Add, Processor, ProcessorGroup  # noqa: F401
from .synths import (DynamicSizeFilteredNoise, FilteredNoise, InHarmonic, MultiAdd,  # noqa: F401
                     MultiInharmonic, SurrogateAdditive)

__all__ = ['core', 'exp_sigmoid', 'exp_tanh', 'Processor', 'ProcessorGroup', 'Add', 'InHarmonic',
           'MultiInharmonic', 'SurrogateAdditive', 'MultiAdd', 'FilteredNoise', 'DynamicSizeFilteredNoise', 'NoiseBandNetSynth',
           'FilterBank', 'Reverb',
           'FeedbackDelayNetwork', 'FeedbackDelayNetworkApply', 'fdn_impulse_response', 'MultiInstrumentReverb',
           'MultiInstrumentFeedbackDelayReverb', 'polyphonic_dag',
           'Parallelizer', 'CapturedGroup', 'NativeGroup', 'MIDIRoll2Conditioning', 'ensure_sequence_length', 'roll_to_conditioning']
