// oscillator.hip, part 1: the template instances of osc_kernel / osc_prepass_fused_kernel with 3 and 4 oscillators per
// lane (see the head of oscillator.hip: one source, four translation units that build in parallel).
#define DDSPP_OSC_PART 1
#include "oscillator.hip"
