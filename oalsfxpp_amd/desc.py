"""ctypes mirror of include/oalsfx_desc.h and of the property structs of include/oalsfxpp.h.

Pure data definitions: no library is loaded here.  Field names and order follow the C
headers exactly; `tests/test_abi_layout.py` checks the sizes against the compiled library.
"""
import ctypes as C

MAX_CHANNELS = 8
MAX_SLOTS = 4
EFFECT_CHANNELS = 4
MAX_CHUNK = 2048

# oalsfxpp::EffectType (reference src/oalsfxpp.h:48-62)
(NULL, CHORUS, COMPRESSOR, DEDICATED_DIALOG, DEDICATED_LFE, DISTORTION, ECHO, EQUALIZER, FLANGER,
 RING_MODULATOR, REVERB, EAX_REVERB) = range(12)
EFFECT_NAMES = ["null", "chorus", "compressor", "dedicated_dialog", "dedicated_low_frequency", "distortion",
                "echo", "equalizer", "flanger", "ring_modulator", "reverb", "eax_reverb"]

# oalsfxpp::ChannelFormat (reference src/oalsfxpp.h:36-46)
FMT_NONE, FMT_MONO, FMT_STEREO, FMT_QUAD, FMT_5POINT1, FMT_5POINT1_REAR, FMT_6POINT1, FMT_7POINT1 = range(8)
FORMAT_CHANNELS = {FMT_MONO: 1, FMT_STEREO: 2, FMT_QUAD: 4, FMT_5POINT1: 6, FMT_5POINT1_REAR: 6, FMT_6POINT1: 7, FMT_7POINT1: 8}

f32, i32, u32 = C.c_float, C.c_int32, C.c_uint32


class _Struct(C.Structure):
    def as_dict(self):
        def conv(v):
            if isinstance(v, _Struct):
                return v.as_dict()
            if isinstance(v, C.Array):
                return [conv(x) for x in v]
            return v
        return {name: conv(getattr(self, name)) for name, *_ in self._fields_}

    def raw(self):
        return bytes(self)


class Biquad(_Struct):
    _fields_ = [("b0", f32), ("b1", f32), ("b2", f32), ("a1", f32), ("a2", f32)]


class Hist(_Struct):
    _fields_ = [("x", f32 * 2), ("y", f32 * 2)]


Gains8 = f32 * MAX_CHANNELS


class SendParams(_Struct):
    _fields_ = [("filter_type", i32), ("out_channels", i32), ("lp", Biquad), ("hp", Biquad), ("gains", Gains8 * MAX_CHANNELS)]


class SourceParams(_Struct):
    _fields_ = [("direct", SendParams), ("aux", SendParams * MAX_SLOTS)]


class SourceState(_Struct):
    _fields_ = [("lp", (Hist * MAX_CHANNELS) * (1 + MAX_SLOTS)), ("hp", (Hist * MAX_CHANNELS) * (1 + MAX_SLOTS))]


class ModDelayParams(_Struct):
    _fields_ = [("waveform", i32), ("delay", i32), ("depth", f32), ("feedback", f32), ("lfo_range", i32),
                ("lfo_scale", f32), ("lfo_disp", i32), ("ring_len", i32), ("gains", Gains8 * 2)]


class ModDelayState(_Struct):
    _fields_ = [("offset", i32)]


class CompressorParams(_Struct):
    _fields_ = [("enabled", i32), ("attack_rate", f32), ("release_rate", f32), ("gains", Gains8 * EFFECT_CHANNELS)]


class CompressorState(_Struct):
    _fields_ = [("gain_control", f32)]


class DedicatedParams(_Struct):
    _fields_ = [("gains", Gains8)]


class DistortionParams(_Struct):
    _fields_ = [("low_pass", Biquad), ("band_pass", Biquad), ("attenuation", f32), ("edge_coeff", f32), ("gains", Gains8)]


class DistortionState(_Struct):
    _fields_ = [("low_pass", Hist), ("band_pass", Hist)]


class EchoParams(_Struct):
    _fields_ = [("tap1", i32), ("tap2", i32), ("feed_gain", f32), ("ring_len", i32), ("filter", Biquad), ("gains", Gains8 * 2)]


class EchoState(_Struct):
    _fields_ = [("offset", i32), ("filter", Hist)]


class EqualizerParams(_Struct):
    _fields_ = [("band", Biquad * 4), ("gains", Gains8 * EFFECT_CHANNELS)]


class EqualizerState(_Struct):
    _fields_ = [("hist", (Hist * EFFECT_CHANNELS) * 4)]


class RingModParams(_Struct):
    _fields_ = [("waveform", i32), ("step", i32), ("filter", Biquad), ("gains", Gains8 * EFFECT_CHANNELS)]


class RingModState(_Struct):
    _fields_ = [("index", i32), ("hist", Hist * EFFECT_CHANNELS)]


class ReverbParams(_Struct):
    _fields_ = [("is_eax", i32), ("lp", Biquad), ("hp", Biquad), ("early_tap", i32 * 4), ("early_tap_coeff", f32 * 4),
                ("late_feed_tap", i32), ("late_tap", i32 * 4), ("ap_feed_coeff", f32), ("mix_x", f32), ("mix_y", f32),
                ("early_ap_off", i32 * 4), ("early_line_off", i32 * 4), ("early_line_coeff", f32 * 4),
                ("mod_range", i32), ("mod_depth", f32), ("mod_coeff", f32), ("density_gain", f32),
                ("late_line_off", i32 * 4), ("late_ap_off", i32 * 4), ("t60_lf", (f32 * 3) * 4), ("t60_hf", (f32 * 3) * 4),
                ("t60_mid", f32 * 4), ("early_pan", Gains8 * 4), ("late_pan", Gains8 * 4), ("ring_len", i32 * 5),
                ("ring_off", i32 * 5)]


class ReverbState(_Struct):
    _fields_ = [("lp", Hist * 4), ("hp", Hist * 4), ("t60", ((f32 * 2) * 2) * 4), ("cur_early_tap", i32 * 4),
                ("cur_early_ap_off", i32 * 4), ("cur_early_line_off", i32 * 4), ("cur_late_tap", i32 * 4),
                ("cur_late_ap_off", i32 * 4), ("cur_late_line_off", i32 * 4), ("mod_index", i32), ("mod_range", i32),
                ("mod_filter", f32), ("fade_count", i32), ("offset", i32), ("early_cur_gain", Gains8 * 4),
                ("late_cur_gain", Gains8 * 4)]


class _SlotParamsU(C.Union):
    _fields_ = [("moddelay", ModDelayParams), ("compressor", CompressorParams), ("dedicated", DedicatedParams),
                ("distortion", DistortionParams), ("echo", EchoParams), ("equalizer", EqualizerParams),
                ("ringmod", RingModParams), ("reverb", ReverbParams)]


class SlotParams(C.Structure):
    _fields_ = [("type", i32), ("update_seq", u32), ("u", _SlotParamsU)]


class _SlotStateU(C.Union):
    _fields_ = [("moddelay", ModDelayState), ("compressor", CompressorState), ("distortion", DistortionState),
                ("echo", EchoState), ("equalizer", EqualizerState), ("ringmod", RingModState), ("reverb", ReverbState)]


class SlotState(C.Structure):
    _fields_ = [("seen_seq", u32), ("u", _SlotStateU)]


# member of the params / state unions that is live for each effect type
PARAMS_MEMBER = {CHORUS: "moddelay", FLANGER: "moddelay", COMPRESSOR: "compressor", DEDICATED_DIALOG: "dedicated",
                 DEDICATED_LFE: "dedicated", DISTORTION: "distortion", ECHO: "echo", EQUALIZER: "equalizer",
                 RING_MODULATOR: "ringmod", REVERB: "reverb", EAX_REVERB: "reverb"}
STATE_MEMBER = {CHORUS: "moddelay", FLANGER: "moddelay", COMPRESSOR: "compressor", DISTORTION: "distortion", ECHO: "echo",
                EQUALIZER: "equalizer", RING_MODULATOR: "ringmod", REVERB: "reverb", EAX_REVERB: "reverb"}


# ---- property structs of the public API (include/oalsfxpp.h; reference src/oalsfxpp.h:65-581) ----
class ChorusProps(_Struct):  # also Flanger
    _fields_ = [("waveform", i32), ("phase", i32), ("rate", f32), ("depth", f32), ("feedback", f32), ("delay", f32)]


class CompressorProps(_Struct):
    _fields_ = [("on_off", C.c_bool)]


class DedicatedProps(_Struct):
    _fields_ = [("gain", f32)]


class DistortionProps(_Struct):
    _fields_ = [("edge", f32), ("gain", f32), ("low_pass_cutoff", f32), ("eq_center", f32), ("eq_bandwidth", f32)]


class EchoProps(_Struct):
    _fields_ = [("delay", f32), ("lr_delay", f32), ("damping", f32), ("feedback", f32), ("spread", f32)]


class EqualizerProps(_Struct):
    _fields_ = [("low_cutoff", f32), ("low_gain", f32), ("mid1_center", f32), ("mid1_gain", f32), ("mid1_width", f32),
                ("mid2_center", f32), ("mid2_gain", f32), ("mid2_width", f32), ("high_cutoff", f32), ("high_gain", f32)]


class ReverbProps(_Struct):
    _fields_ = [("density", f32), ("diffusion", f32), ("gain", f32), ("gain_hf", f32), ("gain_lf", f32), ("decay_time", f32),
                ("decay_hf_ratio", f32), ("decay_lf_ratio", f32), ("reflections_gain", f32), ("reflections_delay", f32),
                ("reflections_pan", f32 * 3), ("late_reverb_gain", f32), ("late_reverb_delay", f32), ("late_reverb_pan", f32 * 3),
                ("echo_time", f32), ("echo_depth", f32), ("modulation_time", f32), ("modulation_depth", f32),
                ("air_absorption_gain_hf", f32), ("hf_reference", f32), ("lf_reference", f32), ("room_rolloff_factor", f32),
                ("decay_hf_limit", C.c_bool)]


class RingModulatorProps(_Struct):
    _fields_ = [("frequency", f32), ("high_pass_cutoff", f32), ("waveform", i32)]


class EffectPropsU(C.Union):
    _fields_ = [("chorus", ChorusProps), ("compressor", CompressorProps), ("dedicated", DedicatedProps),
                ("distortion", DistortionProps), ("echo", EchoProps), ("equalizer", EqualizerProps), ("flanger", ChorusProps),
                ("reverb", ReverbProps), ("ring_modulator", RingModulatorProps), ("raw", C.c_ubyte * 108)]


class Effect(C.Structure):
    """Mirror of oalsfxpp::Effect / oalsfx_effect: 4-byte type tag + 108-byte props union."""
    _fields_ = [("type", i32), ("props", EffectPropsU)]


class SendProps(_Struct):
    _fields_ = [("gain", f32), ("gain_hf", f32), ("gain_lf", f32)]


PROPS_MEMBER = {CHORUS: "chorus", COMPRESSOR: "compressor", DEDICATED_DIALOG: "dedicated", DEDICATED_LFE: "dedicated",
                DISTORTION: "distortion", ECHO: "echo", EQUALIZER: "equalizer", FLANGER: "flanger",
                RING_MODULATOR: "ring_modulator", REVERB: "reverb", EAX_REVERB: "reverb"}

assert C.sizeof(Effect) == 112 and C.sizeof(ReverbProps) == 108 and C.sizeof(SendProps) == 12
