/*
 * oalsfx_desc.h -- plain-data descriptors shared by the host update path, the HIP
 * kernels, the CPU oracle and the tests.
 *
 * One "instance" is what the reference calls an `oalsfxpp::Api` object
 * (reference: src/oalsfxpp.cpp:2820-2905): one source, one direct send, up to
 * four auxiliary sends each feeding one effect slot.
 *
 * The host update path (parameter side, stays on the CPU; reference
 * `EffectState::update` / `calc_panning_and_filters`) produces the *_params
 * structs.  The process path (reference `EffectState::process` bodies and the
 * `Api::Impl::mix_data` slot loop, src/oalsfxpp.cpp:2984-3037) consumes them and
 * owns the *_state structs and the delay-line rings.
 *
 * Everything here is 32-bit words only (int32_t / uint32_t / float), no pointers
 * and no padding surprises, so the same bytes are valid on host and device.
 */
#ifndef OALSFX_DESC_H
#define OALSFX_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OALSFX_MAX_CHANNELS 8      /* reference: max_channels, src/oalsfxpp.cpp:44 */
#define OALSFX_MAX_SLOTS 4         /* reference: max_effects, src/oalsfxpp.cpp:47 */
#define OALSFX_EFFECT_CHANNELS 4   /* reference: max_effect_channels, src/oalsfxpp.cpp:49 */
#define OALSFX_MAX_CHUNK 2048      /* reference: max_sample_buffer_size, src/oalsfxpp.cpp:68 */
#define OALSFX_SILENCE_GAIN 0.00001f /* reference: silence_threshold_gain, src/oalsfxpp.cpp:56 */

/* Effect type numbering == oalsfxpp::EffectType (reference src/oalsfxpp.h:48-62). */
enum {
    OALSFX_NULL = 0,
    OALSFX_CHORUS = 1,
    OALSFX_COMPRESSOR = 2,
    OALSFX_DEDICATED_DIALOG = 3,
    OALSFX_DEDICATED_LFE = 4,
    OALSFX_DISTORTION = 5,
    OALSFX_ECHO = 6,
    OALSFX_EQUALIZER = 7,
    OALSFX_FLANGER = 8,
    OALSFX_RING_MODULATOR = 9,
    OALSFX_REVERB = 10,
    OALSFX_EAX_REVERB = 11,
    OALSFX_TYPE_COUNT = 12
};

/* Channel format numbering == oalsfxpp::ChannelFormat (reference src/oalsfxpp.h:36-46). */
enum {
    OALSFX_FMT_NONE = 0,
    OALSFX_FMT_MONO = 1,
    OALSFX_FMT_STEREO = 2,
    OALSFX_FMT_QUAD = 3,
    OALSFX_FMT_5POINT1 = 4,
    OALSFX_FMT_5POINT1_REAR = 5,
    OALSFX_FMT_6POINT1 = 6,
    OALSFX_FMT_7POINT1 = 7
};

/* Send filter selection == reference ActiveFilters (src/oalsfxpp.cpp:106-112). */
enum { OALSFX_AF_NONE = 0, OALSFX_AF_LOW_PASS = 1, OALSFX_AF_HIGH_PASS = 2, OALSFX_AF_BAND_PASS = 3 };

/* Biquad coefficients, a0 already divided out (reference FilterState members b0_..a2_,
 * src/oalsfxpp.cpp:833-840). */
typedef struct { float b0, b1, b2, a1, a2; } oalsfx_biquad_t;

/* Biquad history: last two inputs and outputs (reference FilterState::x_, y_,
 * src/oalsfxpp.cpp:830-831). Index 0 is the most recent sample. */
typedef struct { float x[2]; float y[2]; } oalsfx_hist_t;

/* ---------------------------------------------------------------------------
 * Source sends (reference Source::Send, src/oalsfxpp.cpp:1093-1152 and
 * calc_panning_and_filters, src/oalsfxpp.cpp:3172-3346)
 * ------------------------------------------------------------------------- */
typedef struct {
    int32_t filter_type;                 /* OALSFX_AF_* */
    int32_t out_channels;                /* direct: device channels; aux: 4, or 0 when the slot is null */
    oalsfx_biquad_t lp;                  /* "low_pass_"  (a high-shelf) */
    oalsfx_biquad_t hp;                  /* "high_pass_" (a low-shelf)  */
    float gains[OALSFX_MAX_CHANNELS][OALSFX_MAX_CHANNELS]; /* [input channel][output] target gains */
} oalsfx_send_params;

typedef struct {
    oalsfx_send_params direct;
    oalsfx_send_params aux[OALSFX_MAX_SLOTS];
} oalsfx_source_params;

typedef struct {
    oalsfx_hist_t lp[1 + OALSFX_MAX_SLOTS][OALSFX_MAX_CHANNELS]; /* [0] direct, [1+i] aux i */
    oalsfx_hist_t hp[1 + OALSFX_MAX_SLOTS][OALSFX_MAX_CHANNELS];
} oalsfx_source_state;

/* ---------------------------------------------------------------------------
 * Per-effect parameters (host-written) and state (process-path-owned)
 * ------------------------------------------------------------------------- */

/* Chorus and flanger share one algorithm (reference src/oalsfxpp.cpp:3972-4277,
 * 5243-5553). */
typedef struct {
    int32_t waveform;        /* 0 sinusoid, 1 triangle */
    int32_t delay;           /* samples */
    float depth;             /* samples */
    float feedback;
    int32_t lfo_range;
    float lfo_scale;
    int32_t lfo_disp;
    int32_t ring_len;        /* power of two, per side */
    float gains[2][OALSFX_MAX_CHANNELS]; /* left / right side gains */
} oalsfx_moddelay_params;

typedef struct { int32_t offset; } oalsfx_moddelay_state;

/* Compressor (reference src/oalsfxpp.cpp:4286-4468). */
typedef struct {
    int32_t enabled;
    float attack_rate;
    float release_rate;
    float gains[OALSFX_EFFECT_CHANNELS][OALSFX_MAX_CHANNELS];
} oalsfx_compressor_params;

typedef struct { float gain_control; } oalsfx_compressor_state;

/* Dedicated dialog / LFE (reference src/oalsfxpp.cpp:4477-4581). */
typedef struct { float gains[OALSFX_MAX_CHANNELS]; } oalsfx_dedicated_params;

/* Distortion (reference src/oalsfxpp.cpp:4590-4762). */
typedef struct {
    oalsfx_biquad_t low_pass;
    oalsfx_biquad_t band_pass;
    float attenuation;
    float edge_coeff;
    float gains[OALSFX_MAX_CHANNELS];
} oalsfx_distortion_params;

typedef struct { oalsfx_hist_t low_pass; oalsfx_hist_t band_pass; } oalsfx_distortion_state;

/* Echo (reference src/oalsfxpp.cpp:4771-4985). */
typedef struct {
    int32_t tap1, tap2;      /* samples */
    float feed_gain;
    int32_t ring_len;        /* power of two */
    oalsfx_biquad_t filter;
    float gains[2][OALSFX_MAX_CHANNELS];
} oalsfx_echo_params;

typedef struct { int32_t offset; oalsfx_hist_t filter; } oalsfx_echo_state;

/* Equalizer (reference src/oalsfxpp.cpp:5034-5232). band 0 low shelf, 1 and 2
 * peaking, 3 high shelf. */
typedef struct {
    oalsfx_biquad_t band[4];
    float gains[OALSFX_EFFECT_CHANNELS][OALSFX_MAX_CHANNELS];
} oalsfx_equalizer_params;

typedef struct { oalsfx_hist_t hist[4][OALSFX_EFFECT_CHANNELS]; /* [band][channel] */ } oalsfx_equalizer_state;

/* Ring modulator (reference src/oalsfxpp.cpp:5556-5785). */
typedef struct {
    int32_t waveform;        /* 0 sin, 1 saw, 2 square */
    int32_t step;
    oalsfx_biquad_t filter;  /* b0=a, b1=-a, b2=0, a1=-a, a2=0 */
    float gains[OALSFX_EFFECT_CHANNELS][OALSFX_MAX_CHANNELS];
} oalsfx_ringmod_params;

typedef struct { int32_t index; oalsfx_hist_t hist[OALSFX_EFFECT_CHANNELS]; } oalsfx_ringmod_state;

/* Reverb and EAX reverb (reference src/oalsfxpp.cpp:5799-7904).
 * Ring numbering: 0 main delay, 1 early all-pass, 2 early line, 3 late all-pass,
 * 4 late line.  Unlike the reference (which interleaves the four lines inside a
 * frame, src/oalsfxpp.cpp:6195) every line of every ring is a contiguous ring of
 * ring_len[r] floats: line j of ring r starts ring_off[r] + j*ring_len[r] floats
 * into the instance's ring slab. */
#define OALSFX_RV_MAIN 0
#define OALSFX_RV_EARLY_AP 1
#define OALSFX_RV_EARLY_LINE 2
#define OALSFX_RV_LATE_AP 3
#define OALSFX_RV_LATE_LINE 4
#define OALSFX_RV_FADE_SAMPLES 128   /* reference fade_samples, src/oalsfxpp.cpp:6187 */
#define OALSFX_RV_MAX_UPDATE 256     /* reference max_update_samples, src/oalsfxpp.cpp:6181 */

typedef struct {
    int32_t is_eax;
    oalsfx_biquad_t lp;                  /* filters_[*].lp_ (same for the 4 lines) */
    oalsfx_biquad_t hp;                  /* filters_[*].hp_ (EAX only) */
    int32_t early_tap[4];                /* early_delay_taps_[j][1] */
    float early_tap_coeff[4];            /* early_delay_coeffs_ */
    int32_t late_feed_tap;
    int32_t late_tap[4];                 /* late_delay_taps_[j][1] */
    float ap_feed_coeff;
    float mix_x, mix_y;
    int32_t early_ap_off[4];             /* early_.vec_ap_.offsets_[j][1] */
    int32_t early_line_off[4];           /* early_.offsets_[j][1] */
    float early_line_coeff[4];           /* early_.coeffs_ */
    int32_t mod_range;
    float mod_depth;
    float mod_coeff;
    float density_gain;
    int32_t late_line_off[4];            /* late_.offsets_[j][1] */
    int32_t late_ap_off[4];              /* late_.vec_ap_.offsets_[j][1] */
    float t60_lf[4][3];
    float t60_hf[4][3];
    float t60_mid[4];
    float early_pan[4][OALSFX_MAX_CHANNELS];
    float late_pan[4][OALSFX_MAX_CHANNELS];
    int32_t ring_len[5];
    int32_t ring_off[5];
} oalsfx_reverb_params;

typedef struct {
    oalsfx_hist_t lp[4];
    oalsfx_hist_t hp[4];
    float t60[4][2][2];                  /* [line][section][last_in,last_out] */
    /* the "current" taps, index [..][0] in the reference; copied from the target
     * taps when a cross-fade completes (src/oalsfxpp.cpp:6128-6136) */
    int32_t cur_early_tap[4];
    int32_t cur_early_ap_off[4];
    int32_t cur_early_line_off[4];
    int32_t cur_late_tap[4];
    int32_t cur_late_ap_off[4];
    int32_t cur_late_line_off[4];
    int32_t mod_index;
    int32_t mod_range;                   /* range the index is currently scaled to (starts at 1) */
    float mod_filter;
    int32_t fade_count;
    int32_t offset;
    float early_cur_gain[4][OALSFX_MAX_CHANNELS];
    float late_cur_gain[4][OALSFX_MAX_CHANNELS];
} oalsfx_reverb_state;

/* Placement of the five reverb rings inside an instance's slab: longest ring first (stable), four
 * lines each.  All lengths are powers of two, so every line then starts at a multiple of its own
 * length and a kernel can form "line base | (position & (len-1))" with one AND-OR. */
static inline void oalsfx_reverb_place_rings(const int32_t len[5], int32_t off[5])
{
    int32_t order[5] = {0, 1, 2, 3, 4};
    int32_t at = 0;
    for (int i = 1; i < 5; ++i)
        for (int k = i; k > 0 && len[order[k]] > len[order[k - 1]]; --k) {
            const int32_t t = order[k]; order[k] = order[k - 1]; order[k - 1] = t;
        }
    for (int i = 0; i < 5; ++i) {
        off[order[i]] = at;
        at += 4 * len[order[i]];
    }
}

/* ---------------------------------------------------------------------------
 * One effect slot of one instance
 * ------------------------------------------------------------------------- */
typedef struct {
    int32_t type;        /* OALSFX_* effect type */
    uint32_t update_seq; /* incremented each time the reference would call EffectState::update */
    union {
        oalsfx_moddelay_params moddelay;      /* chorus, flanger */
        oalsfx_compressor_params compressor;
        oalsfx_dedicated_params dedicated;
        oalsfx_distortion_params distortion;
        oalsfx_echo_params echo;
        oalsfx_equalizer_params equalizer;
        oalsfx_ringmod_params ringmod;
        oalsfx_reverb_params reverb;
    } u;
} oalsfx_slot_params;

typedef struct {
    uint32_t seen_seq;   /* last update_seq the process path has folded into the state */
    union {
        oalsfx_moddelay_state moddelay;
        oalsfx_compressor_state compressor;
        oalsfx_distortion_state distortion;
        oalsfx_echo_state echo;
        oalsfx_equalizer_state equalizer;
        oalsfx_ringmod_state ringmod;
        oalsfx_reverb_state reverb;
    } u;
} oalsfx_slot_state;

#ifdef __cplusplus
}
#endif

#endif /* OALSFX_DESC_H */
