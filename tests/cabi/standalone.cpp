// A caller of libddspp.so that knows NOTHING but include/ddspp.h and the HIP runtime: no Python, no torch.
// It renders the two committed golden cases (tests/golden: BASELINE config 1 and the down-sized config 2) from raw
// float32 files and writes the audio back; tests/test_gpu_cabi_standalone.py exports the inputs, runs this binary on
// the GPU box and compares its output with the golden audio.  This is the shape of the binding a maintainer of the
// reference would write: table builders -> get_controls -> synthesis -> mix -> reverb.
//
//   standalone c1 <dir>      reads  raw_amplitudes.f32 raw_harmonic_distribution.f32 raw_inharm_coef.f32 raw_f0_hz.f32
//   standalone c2 <dir>      reads  <key>_<voice>.f32 for the five controls, noise_<voice>.f32, reverb_ir.f32
//   both write <dir>/audio.f32 (c2 also dry.f32).  Dimensions come from <dir>/dims.txt (one integer per line).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "ddspp.h"

#define CK(x)                                                                  \
    do {                                                                       \
        const int _rc = (x);                                                   \
        if (_rc != 0) {                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #x, _rc, ddspp_last_error());    \
            exit(2);                                                           \
        }                                                                      \
    } while (0)
#define HK(x)                                                                  \
    do {                                                                       \
        const hipError_t _e = (x);                                             \
        if (_e != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e));            \
            exit(3);                                                           \
        }                                                                      \
    } while (0)

static std::vector<float> read_f32(const std::string& path, size_t n) {
    std::vector<float> v(n);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f || fread(v.data(), sizeof(float), n, f) != n) {
        fprintf(stderr, "cannot read %zu floats from %s\n", n, path.c_str());
        exit(4);
    }
    fclose(f);
    return v;
}

static void write_f32(const std::string& path, const std::vector<float>& v) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(v.data(), sizeof(float), v.size(), f) != v.size()) exit(5);
    fclose(f);
}

template <class T>
static T* upload(const std::vector<T>& h) {
    T* d = nullptr;
    HK(hipMalloc(&d, h.size() * sizeof(T) + 16));
    HK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static float* dalloc(size_t n) {
    float* d = nullptr;
    HK(hipMalloc(&d, n * sizeof(float) + 16));
    return d;
}

static std::vector<float> download(const float* d, size_t n) {
    std::vector<float> h(n);
    HK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
    return h;
}

struct Tables {
    float *wlin, *whann;
};

// wlin / whann of ddspp_harmonic_synthesis from the library's own builders
static Tables synthesis_tables(int T, int U) {
    const int N = T * U;
    std::vector<int> lo(N), hi(N);
    std::vector<float> w(N), hann(2 * U);
    int aligned = 0;
    CK(ddspp_resample_tables_host(T, N, 0, lo.data(), hi.data(), w.data(), &aligned));
    if (!aligned) {
        fprintf(stderr, "frame / sample rates not aligned\n");
        exit(6);
    }
    CK(ddspp_hann_window_host(2 * U, hann.data()));
    return Tables{upload(w), upload(hann)};
}

// MultiInharmonic(inference=True)(amplitudes, harmonic_distribution, inharm_coef, f0_hz) for rows [R, T, .]
static float* multi_inharmonic(const float* amp, const float* hd, const float* inh, const float* f0, int R, int T, int H,
                               int S, int U, float sr, const Tables& tb) {
    const size_t frames = (size_t)R * T;
    float *amp_c = dalloc(frames), *hd_c = dalloc(frames * H), *sh_c = dalloc(frames * H);
    CK(ddspp_inharmonic_controls(amp, hd, inh, f0, amp_c, hd_c, sh_c, nullptr, R, T, H, S, sr, 20.0f, DDSPP_SCALE_EXP_SIGMOID,
                                 10.0f, 2.0f, 1e-7f, 1.0f, 1, 1, nullptr));
    const int N = T * U;
    float* audio = dalloc((size_t)R * N);
    const size_t wsb = ddspp_osc_workspace_bytes(R, N, S * H);
    void* ws = nullptr;
    HK(hipMalloc(&ws, wsb + 256));
    CK(ddspp_harmonic_synthesis(f0, amp_c, hd_c, sh_c, tb.wlin, tb.whann, audio, R, T, S, H, U, sr, 1, 0, ws, wsb, nullptr));
    HK(hipDeviceSynchronize());
    HK(hipFree(ws)); HK(hipFree(amp_c)); HK(hipFree(hd_c)); HK(hipFree(sh_c));
    return audio;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s c1|c2 <dir>\n", argv[0]);
        return 1;
    }
    const std::string which = argv[1], dir = std::string(argv[2]) + "/";
    std::vector<int> dims;
    {
        FILE* f = fopen((dir + "dims.txt").c_str(), "r");
        int v;
        while (f && fscanf(f, "%d", &v) == 1) dims.push_back(v);
        if (f) fclose(f);
    }
    printf("libddspp %d for %s\n", ddspp_version(), ddspp_target_arch());
    if (which == "c1") {
        if (dims.size() < 4) return 1;
        const int T = dims[0], H = dims[1], sr = dims[2], U = dims[3], N = T * U;
        const Tables tb = synthesis_tables(T, U);
        float* amp = upload(read_f32(dir + "raw_amplitudes.f32", T));
        float* hd = upload(read_f32(dir + "raw_harmonic_distribution.f32", (size_t)T * H));
        float* inh = upload(read_f32(dir + "raw_inharm_coef.f32", T));
        float* f0 = upload(read_f32(dir + "raw_f0_hz.f32", T));
        float* audio = multi_inharmonic(amp, hd, inh, f0, 1, T, H, 1, U, (float)sr, tb);
        write_f32(dir + "audio.f32", download(audio, N));
        return 0;
    }
    if (which == "c2") {
        if (dims.size() < 8) return 1;
        const int P = dims[0], T = dims[1], H = dims[2], K = dims[3], S = dims[4], sr = dims[5], U = dims[6], L = dims[7];
        const int N = T * U, B = 1;
        const Tables tb = synthesis_tables(T, U);
        // FilteredNoise tables: even/odd design when the shape has one, the generic matrix otherwise
        int Lw = 0, NJ = 0;
        CK(ddspp_fir_tables_shape(K, 257, &Lw, &NJ));
        float *CE = nullptr, *CO = nullptr, *we = nullptr, *wo = nullptr, *M = nullptr;
        int *idx = nullptr, *uniq = nullptr, *mirror = nullptr, n_uniq = 0;
        if (NJ > 0) {
            std::vector<float> hce((size_t)(K / 2) * NJ), hco(hce.size()), hwe(NJ * 4), hwo(NJ * 4);
            std::vector<int> hidx(NJ * 4);
            CK(ddspp_fir_eo_tables_host(K, 257, hce.data(), hco.data(), hidx.data(), hwe.data(), hwo.data()));
            CE = upload(hce); CO = upload(hco); idx = upload(hidx); we = upload(hwe); wo = upload(hwo);
        } else {
            std::vector<float> hm((size_t)K * Lw);
            std::vector<int> hu(Lw), hmi(Lw);
            CK(ddspp_fir_matrix_host(K, 257, 0, hm.data(), hu.data(), hmi.data(), &n_uniq));
            M = upload(hm); uniq = upload(hu); mirror = upload(hmi);
        }
        float* additive = dalloc((size_t)B * P * N);          // [B, P, N]
        float* noise_sig = dalloc((size_t)B * P * N);
        for (int v = 0; v < P; ++v) {
            const std::string sv = "_" + std::to_string(v) + ".f32";
            float* amp = upload(read_f32(dir + "amplitudes" + sv, T));
            float* hd = upload(read_f32(dir + "harmonic_distribution" + sv, (size_t)T * H));
            float* inh = upload(read_f32(dir + "inharm_coef" + sv, T));
            float* f0 = upload(read_f32(dir + "f0_hz" + sv, (size_t)T * S));
            float* a = multi_inharmonic(amp, hd, inh, f0, B, T, H, S, U, (float)sr, tb);
            HK(hipMemcpy(additive + (size_t)v * N, a, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice));
            // DynamicSizeFilteredNoise: get_controls (scale_fn(magnitudes - 5)) fused into the design, explicit noise
            float* mags = upload(read_f32(dir + "magnitudes" + sv, (size_t)T * K));
            float* z = upload(read_f32(dir + "noise" + sv, N));
            float* ir = dalloc((size_t)T * Lw);
            if (NJ > 0) {
                CK(ddspp_fir_from_magnitudes_eo(mags, CE, CO, idx, we, wo, ir, (size_t)B * T, K, Lw, NJ, DDSPP_SCALE_EXP_SIGMOID,
                                                -5.0f, 10.0f, 2.0f, 1e-7f, 1.0f, nullptr));
            } else {
                float* scaled = dalloc((size_t)T * K);
                CK(ddspp_scale_bias(mags, scaled, (size_t)T * K, -5.0f, DDSPP_SCALE_EXP_SIGMOID, 10.0f, 2.0f, 1e-7f, 1.0f, nullptr));
                CK(ddspp_fir_from_magnitudes(scaled, M, uniq, mirror, n_uniq, ir, (size_t)B * T, K, Lw, nullptr));
            }
            CK(ddspp_time_varying_fir(z, ir, noise_sig + (size_t)v * N, B, N, T, Lw, DDSPP_DELAY_AUTO, nullptr));
        }
        // the add chain of polyphonic_dag.py:28-37, then ddsp.effects.Reverb (dry tap masked, dry added)
        float* dry = dalloc((size_t)B * N);
        CK(ddspp_polyphonic_mix(additive, noise_sig, dry, nullptr, B, P, N, N, 0, nullptr));
        float* rir = upload(read_f32(dir + "reverb_ir.f32", L));
        ddspp_fftconv_plan* plan = nullptr;
        CK(ddspp_fftconv_plan_create(B, 1, N, L, &plan));
        const size_t wsb = ddspp_fftconv_workspace_bytes(plan);
        void* ws = nullptr;
        HK(hipMalloc(&ws, wsb + 256));
        float* wet = dalloc((size_t)B * N);
        CK(ddspp_fftconv_execute(plan, dry, N, rir, wet, N, 0, 1, 1, ws, wsb, nullptr));
        HK(hipDeviceSynchronize());
        write_f32(dir + "dry.f32", download(dry, (size_t)B * N));
        write_f32(dir + "audio.f32", download(wet, (size_t)B * N));
        CK(ddspp_fftconv_plan_destroy(plan));

        // ---- the same segment through the one-call driver: processor_group(features, return_outputs_dict=True) ---------
        auto stacked = [&](const char* key, size_t per_voice) {             // rows [B = 1, P] segment major
            std::vector<float> all;
            for (int v = 0; v < P; ++v) {
                const std::vector<float> x = read_f32(dir + key + "_" + std::to_string(v) + ".f32", per_voice);
                all.insert(all.end(), x.begin(), x.end());
            }
            return upload(all);
        };
        ddspp_group_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.n_segments = B; cfg.n_voices = P; cfg.n_frames = T; cfg.n_harmonics = H; cfg.n_substrings = S; cfg.n_bands = K;
        cfg.upsampling = U; cfg.ir_length = L; cfg.ir_batch = 1; cfg.reverb_add_dry = 1; cfg.voice_major = 0;
        cfg.sample_rate = (float)sr; cfg.min_frequency = 20.0f;
        cfg.scale_kind = DDSPP_SCALE_EXP_SIGMOID; cfg.exponent = 10.0f; cfg.max_value = 2.0f; cfg.threshold = 1e-7f; cfg.gain = 1.0f;
        cfg.normalize_after_nyquist_cut = 1; cfg.normalize_below_nyquist = 1;
        cfg.window_size = 257;
        cfg.noise_scale_kind = DDSPP_SCALE_EXP_SIGMOID; cfg.noise_bias = -5.0f; cfg.noise_exponent = 10.0f;
        cfg.noise_max_value = 2.0f; cfg.noise_threshold = 1e-7f; cfg.noise_gain = 1.0f;
        cfg.delay_compensation = DDSPP_DELAY_AUTO; cfg.resize_rule = 0; cfg.noise_seed = 0;
        ddspp_group* grp = nullptr;
        CK(ddspp_group_create(&cfg, &grp));
        const size_t gws_bytes = ddspp_group_workspace_bytes(grp);
        void* gws = nullptr;
        HK(hipMalloc(&gws, gws_bytes));
        float* g_audio = dalloc((size_t)B * N);
        ddspp_group_outputs go;
        memset(&go, 0, sizeof(go));
        go.dry = dalloc((size_t)B * N);
        go.additive_last = dalloc((size_t)B * N);
        go.noise_last = dalloc((size_t)B * N);
        CK(ddspp_group_run(grp, stacked("amplitudes", T), stacked("harmonic_distribution", (size_t)T * H), stacked("inharm_coef", T),
                           stacked("f0_hz", (size_t)T * S), stacked("magnitudes", (size_t)T * K), rir, stacked("noise", N), g_audio,
                           &go, gws, gws_bytes, nullptr));
        HK(hipDeviceSynchronize());
        write_f32(dir + "audio_group.f32", download(g_audio, (size_t)B * N));
        write_f32(dir + "dry_group.f32", download(go.dry, (size_t)B * N));
        write_f32(dir + "additive_last_group.f32", download(go.additive_last, (size_t)B * N));
        write_f32(dir + "noise_last_group.f32", download(go.noise_last, (size_t)B * N));
        ddspp_group_destroy(grp);
        return 0;
    }
    return 1;
}
