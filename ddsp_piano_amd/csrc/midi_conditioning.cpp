// MIDI piano-roll -> polyphonic conditioning (host side, CPU).
//
// Replaces ddsp_piano/utils/midi_encoders.py:4-104 (MIDIRoll2Conditioning), the step that feeds the control
// networks whose outputs enter the synthesis path (io_utils.py:118-120).  It is a frame-sequential voice
// allocator over 88 x n_synths values per 4 ms frame -- integer-like bookkeeping with a loop-carried state, so
// it stays on the host: a 10 minute file (150 000 frames) takes a few milliseconds here against tens of
// seconds of NumPy calls in the reference.  Arithmetic and tie handling are stated in include/ddspp.h.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "ddspp_common.h"

namespace {

constexpr int kPitches = 88;      // midi_encoders.py:19: pitch_mul = arange(21, 21 + 88)
constexpr int kLowestPitch = 21;

struct State {
    int n = 0;
    int assigner = 0;                    // next free channel, -1 when every channel holds a note
    std::vector<int> reorder;            // conditioning channel -> position in the frame's sorted pitch list
    std::vector<double> assigned;        // pitch held by each channel (0 = free)
};

inline bool contains(const double* v, int n, double x) {
    for (int i = 0; i < n; ++i)
        if (v[i] == x) return true;
    return false;
}

// midi_encoders.py:23-31
void advance_assigner(State& s) {
    const int n = s.n;
    s.assigner = ((s.assigner + 1) % n + n) % n;
    if (!contains(s.assigned.data(), n, 0.0)) {
        s.assigner = -1;
        return;
    }
    while (s.assigned[s.assigner] != 0.0) s.assigner = (s.assigner + 1) % n;
}

inline int wrap(int i, int n) { return i < 0 ? i + n : i; }      // NumPy's negative indexing

// np.sum over 88 strided values: NumPy's pairwise summation reduces n < 128 with eight running sums
template <typename T>
T pairwise88(const T* a, int stride) {
    T r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
    for (int i = 8; i < kPitches; i += 8)
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a[(i + j) * stride];
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

template <typename T>
int run(State& s, const T* roll, int n_frames, T* cond, T* polyphony) {
    const int n = s.n;
    std::vector<double> pitches(n), vel(n), value(kPitches), prev(kPitches);
    std::vector<int> order(kPitches), reorder(n);
    bool consistent = false;             // do the channels hold exactly the current frame's set of pitches?
    for (int t = 0; t < n_frames; ++t) {
        const T* fr = roll + (size_t)t * kPitches * 2;
        polyphony[t] = pairwise88<T>(fr, 2);                                          // :47
        bool unchanged = t > 0;
        for (int k = 0; k < kPitches; ++k) {
            value[k] = (double)(T)((double)fr[2 * k] * (double)(kLowestPitch + k));   // :49 (product rounded to T)
            unchanged = unchanged && value[k] == prev[k];
        }
        if (!unchanged) {
            // :52-55 the n_synths largest values in ascending order; equal values keep ascending key order
            for (int k = 0; k < kPitches; ++k) order[k] = k;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return value[a] < value[b]; });
            for (int c = 0; c < n; ++c) pitches[c] = value[order[kPitches - n + c]];
            prev = value;
        }   // else: same keys down as in the previous frame -> same order, same pitches (the common case)
        for (int c = 0; c < n; ++c) vel[c] = (double)fr[2 * order[kPitches - n + c] + 1];
        // :61-69 same set of pitches as the channels hold: the previous permutation still applies
        auto sets_match = [&]() {
            for (int c = 0; c < n; ++c)
                if (!contains(s.assigned.data(), n, pitches[c]) || !contains(pitches.data(), n, s.assigned[c]))
                    return false;
            return true;
        };
        // unchanged frame: neither side of the comparison moved since it was last evaluated
        const bool same = t > 0 && (unchanged ? consistent : sets_match());
        consistent = same ? true : consistent;
        if (!same) {
            std::fill(reorder.begin(), reorder.end(), 0);                             // :72
            for (int c = 0; c < n; ++c)                                               // :74-79 free finished notes
                if (!contains(pitches.data(), n, s.assigned[c])) {
                    s.assigned[c] = 0.0;
                    if (s.assigner == -1) advance_assigner(s);
                }
            for (int c = 0; c < n; ++c)                                               // :82-85 sustained notes stay
                if (pitches[c] != 0.0 && contains(s.assigned.data(), n, pitches[c])) {
                    int ch = 0;
                    while (s.assigned[ch] != pitches[c]) ++ch;
                    reorder[ch] = c;
                }
            for (int c = 0; c < n; ++c)                                               // :88-92 new notes
                if (!contains(s.assigned.data(), n, pitches[c])) {
                    const int ch = wrap(s.assigner, n);
                    reorder[ch] = c;
                    s.assigned[ch] = pitches[c];
                    advance_assigner(s);
                }
            for (int c = 0; c < n; ++c)                                               // :95-98 silent channels
                if (pitches[c] == 0.0) {
                    reorder[wrap(s.assigner, n)] = c;
                    advance_assigner(s);
                }
            s.reorder = reorder;                                                      // :102
            consistent = sets_match();
        }
        T* out = cond + (size_t)t * n * 2;
        for (int ch = 0; ch < n; ++ch) {                                              // :66-67 / :100-101
            out[2 * ch] = (T)pitches[s.reorder[ch]];
            out[2 * ch + 1] = (T)vel[s.reorder[ch]];
        }
    }
    return DDSPP_OK;
}

}  // namespace

extern "C" {

struct ddspp_midi_state {
    State s;
};

int ddspp_midi_conditioning_reset(ddspp_midi_state* h);

ddspp_midi_state* ddspp_midi_conditioning_create(int n_synths) {
    if (n_synths < 1 || n_synths > kPitches) {
        ddspp_set_error("midi_conditioning_create: n_synths=%d outside 1..%d", n_synths, kPitches);
        return nullptr;
    }
    ddspp_midi_state* h = new (std::nothrow) ddspp_midi_state();
    if (!h) return nullptr;
    h->s.n = n_synths;
    ddspp_midi_conditioning_reset(h);
    return h;
}

void ddspp_midi_conditioning_destroy(ddspp_midi_state* h) { delete h; }

// the state of a freshly constructed MIDIRoll2Conditioning (midi_encoders.py:16-22)
int ddspp_midi_conditioning_reset(ddspp_midi_state* h) {
    DDSPP_REQUIRE(h, "midi_conditioning_reset: null state");
    State& s = h->s;
    s.assigner = 0;
    s.reorder.resize(s.n);
    for (int i = 0; i < s.n; ++i) s.reorder[i] = i;
    s.assigned.assign(s.n, 0.0);
    return DDSPP_OK;
}

int ddspp_midi_conditioning_get_state(const ddspp_midi_state* h, int* assigner, int* reorder,
                                      double* assigned_pitch) {
    DDSPP_REQUIRE(h, "midi_conditioning_get_state: null state");
    if (assigner) *assigner = h->s.assigner;
    if (reorder) std::memcpy(reorder, h->s.reorder.data(), sizeof(int) * h->s.n);
    if (assigned_pitch) std::memcpy(assigned_pitch, h->s.assigned.data(), sizeof(double) * h->s.n);
    return DDSPP_OK;
}

int ddspp_midi_conditioning_run_f64(ddspp_midi_state* h, const double* roll, int n_frames, int n_pitches,
                                    double* conditioning, double* polyphony) {
    DDSPP_REQUIRE(h, "midi_conditioning_run: null state");
    DDSPP_REQUIRE(n_frames >= 0 && (n_frames == 0 || (roll && conditioning && polyphony)),
                  "midi_conditioning_run: null buffer");
    DDSPP_REQUIRE(n_pitches == kPitches, "midi_conditioning_run: roll has %d keys, expected %d (MIDI 21..108)",
                  n_pitches, kPitches);
    return run<double>(h->s, roll, n_frames, conditioning, polyphony);
}

int ddspp_midi_conditioning_run_f32(ddspp_midi_state* h, const float* roll, int n_frames, int n_pitches,
                                    float* conditioning, float* polyphony) {
    DDSPP_REQUIRE(h, "midi_conditioning_run: null state");
    DDSPP_REQUIRE(n_frames >= 0 && (n_frames == 0 || (roll && conditioning && polyphony)),
                  "midi_conditioning_run: null buffer");
    DDSPP_REQUIRE(n_pitches == kPitches, "midi_conditioning_run: roll has %d keys, expected %d (MIDI 21..108)",
                  n_pitches, kPitches);
    return run<float>(h->s, roll, n_frames, conditioning, polyphony);
}

}  // extern "C"
