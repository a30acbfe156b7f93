// oalsfx_wav -- runs a WAV file through one effect on the MI355X backend, through the unchanged oalsfxpp::Api.
//
// Counterpart of the reference's demo program (reference src/oalsfxpp_test.cpp:744-901): same command line
// (`program <src> <dst>`, effect chosen from the same numbered menu on stdin), same sample conversions
// (8/16-bit PCM in: test.cpp:706-741; 16-bit PCM out scaled so that nothing clips: test.cpp:602-636).  The effect
// can also be given as a third argument (menu number or effect name) for scripted use.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "oalsfxpp.h"

namespace {

struct Wav {
    int channels = 0;
    int rate = 0;
    int bits = 0;
    std::vector<float> samples; // interleaved
    std::string error;
};

uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
uint16_t le16(const uint8_t* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

bool read_wav(const std::string& name, Wav& w)
{
    std::ifstream f(name, std::ios::binary);
    if (!f) { w.error = "Failed to open a file \"" + name + "\"."; return false; }
    const std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (d.size() < 12 || std::memcmp(d.data(), "RIFF", 4) != 0 || std::memcmp(d.data() + 8, "WAVE", 4) != 0) { w.error = "Not a WAV stream."; return false; }
    const uint8_t* fmt = nullptr; const uint8_t* data = nullptr;
    size_t fmt_size = 0, data_size = 0;
    for (size_t at = 12; at + 8 <= d.size();) {
        const size_t size = le32(d.data() + at + 4);
        const uint8_t* body = d.data() + at + 8;
        const size_t avail = std::min(size, d.size() - (at + 8));
        if (std::memcmp(d.data() + at, "fmt ", 4) == 0) {
            if (fmt) { w.error = "Multiple format chunks."; return false; }
            fmt = body; fmt_size = avail;
        } else if (std::memcmp(d.data() + at, "data", 4) == 0) {
            if (data) { w.error = "Multiple data chunks."; return false; }
            data = body; data_size = avail;
        }
        at += 8 + size + (size & 1);
    }
    if (!fmt) { w.error = "Format chunk not found."; return false; }
    if (!data) { w.error = "Data chunk not found."; return false; }
    if (fmt_size < 16) { w.error = "Invalid format chunk."; return false; }
    if (le16(fmt) != 1) { w.error = "Expected a PCM codec."; return false; }
    w.channels = le16(fmt + 2);
    w.rate = static_cast<int>(le32(fmt + 4));
    w.bits = le16(fmt + 14);
    if (w.channels < 1 || w.channels > oalsfxpp::Api::get_max_channels()) { w.error = "Channel count is out of range."; return false; }
    if (w.rate < oalsfxpp::Api::get_min_sampling_rate()) { w.error = "Sampling rate is out of range."; return false; }
    if (w.bits != 8 && w.bits != 16) { w.error = "Unsupported bit depth."; return false; }
    const size_t frame_bytes = static_cast<size_t>(w.channels) * (w.bits / 8);
    const size_t total = data_size / frame_bytes * w.channels;
    if (total == 0) { w.error = "No data to read."; return false; }
    w.samples.resize(total);
    if (w.bits == 8) {
        for (size_t i = 0; i < total; ++i) w.samples[i] = (static_cast<int>(data[i]) - 128) / 128.0F;
    } else {
        for (size_t i = 0; i < total; ++i) w.samples[i] = static_cast<int16_t>(le16(data + 2 * i)) / 32768.0F;
    }
    return true;
}

void put16(std::vector<uint8_t>& o, uint16_t v) { o.push_back(v & 0xFF); o.push_back(v >> 8); }
void put32(std::vector<uint8_t>& o, uint32_t v) { put16(o, v & 0xFFFF); put16(o, v >> 16); }

bool write_wav_s16(const std::string& name, const Wav& like, const std::vector<float>& samples, std::string& error)
{
    if (samples.empty()) { error = "No data to write."; return false; }
    // headroom: the loudest sample maps to full scale when anything exceeds [-1, 1]
    float lo = -1.0F, hi = 1.0F;
    for (float s : samples) {
        if (s < lo) lo = s;
        else if (s > hi) hi = s;
    }
    const float scale = 1.0F / std::max(hi, -lo);
    std::vector<uint8_t> o;
    const uint32_t data_bytes = static_cast<uint32_t>(samples.size() * 2);
    o.reserve(44 + data_bytes);
    o.insert(o.end(), {'R', 'I', 'F', 'F'}); put32(o, 36 + data_bytes);
    o.insert(o.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '}); put32(o, 16);
    put16(o, 1); put16(o, static_cast<uint16_t>(like.channels)); put32(o, static_cast<uint32_t>(like.rate));
    put32(o, static_cast<uint32_t>(like.rate) * like.channels * 2); put16(o, static_cast<uint16_t>(like.channels * 2)); put16(o, 16);
    o.insert(o.end(), {'d', 'a', 't', 'a'}); put32(o, data_bytes);
    for (float s : samples) put16(o, static_cast<uint16_t>(static_cast<int16_t>(scale * s * 32767.0F)));
    std::ofstream f(name, std::ios::binary);
    if (!f) { error = "Failed to open a file \"" + name + "\"."; return false; }
    f.write(reinterpret_cast<const char*>(o.data()), static_cast<std::streamsize>(o.size()));
    if (!f) { error = "Failed to write data."; return false; }
    return true;
}

struct MenuEntry { const char* label; const char* name; oalsfxpp::EffectType type; };
const MenuEntry kMenu[] = {
    {"EAX Reverb", "eax_reverb", oalsfxpp::EffectType::eax_reverb},
    {"Reverb", "reverb", oalsfxpp::EffectType::reverb},
    {"Chorus", "chorus", oalsfxpp::EffectType::chorus},
    {"Compressor", "compressor", oalsfxpp::EffectType::compressor},
    {"Dedicated (dialog)", "dedicated_dialog", oalsfxpp::EffectType::dedicated_dialog},
    {"Dedicated (low frequency)", "dedicated_low_frequency", oalsfxpp::EffectType::dedicated_low_frequency},
    {"Distortion", "distortion", oalsfxpp::EffectType::distortion},
    {"Echo", "echo", oalsfxpp::EffectType::echo},
    {"Equalizer", "equalizer", oalsfxpp::EffectType::equalizer},
    {"Flanger", "flanger", oalsfxpp::EffectType::flanger},
    {"Ring modulator", "ring_modulator", oalsfxpp::EffectType::ring_modulator},
    {"Null", "null", oalsfxpp::EffectType::null},
};
constexpr int kMenuSize = sizeof(kMenu) / sizeof(kMenu[0]);

int menu_index(const std::string& text)
{
    for (int i = 0; i < kMenuSize; ++i)
        if (text == kMenu[i].name) return i;
    char* end = nullptr;
    const long v = std::strtol(text.c_str(), &end, 10);
    if (end != text.c_str() && *end == '\0' && v >= 1 && v <= kMenuSize) return static_cast<int>(v - 1);
    return -1;
}

} // namespace

int main(int argc, char* argv[])
{
    if (argc != 3 && argc != 4) {
        std::cout << "Usage:" << std::endl;
        std::cout << "program <src_file_name> <dst_file_name> [effect number or name]" << std::endl;
        return 1;
    }
    Wav wav;
    if (!read_wav(argv[1], wav)) { std::cout << wav.error << std::endl; return 2; }

    oalsfxpp::Api api;
    if (!api.initialize(oalsfxpp::Api::channel_count_to_channel_format(wav.channels), wav.rate, 1)) {
        std::cout << api.get_error_message() << std::endl;
        return 2;
    }

    int choice = -1;
    if (argc == 4) {
        choice = menu_index(argv[3]);
        if (choice < 0) { std::cout << "Unknown effect \"" << argv[3] << "\"." << std::endl; return 1; }
    } else {
        for (int i = 0; i < kMenuSize; ++i) std::cout << (i + 1) << ". " << kMenu[i].label << "\n";
        std::cout << std::endl;
        while (choice < 0) {
            std::cout << "Enter effect number: ";
            std::string line;
            if (!(std::cin >> line)) return 2;
            choice = menu_index(line);
        }
    }
    api.set_effect_type(0, kMenu[choice].type);
    api.apply_changes();

    const int frames = static_cast<int>(wav.samples.size() / wav.channels);
    std::vector<float> out(wav.samples.size());
    if (!api.mix(frames, wav.samples.data(), out.data())) { std::cout << api.get_error_message() << std::endl; return 2; }
    std::string error;
    if (!write_wav_s16(argv[2], wav, out, error)) { std::cout << error << std::endl; return 2; }
    return 0;
}
