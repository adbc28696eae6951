// fw_oracle.cpp — CPU ORACLE (test infrastructure only; see fw_oracle.hpp header).
// Every function cites the reference file:line it restates.
#include "fw_oracle.hpp"

#include <atomic>
#include <chrono>
#include <thread>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <tuple>

namespace fwo {

// ================================================================= core/param/smoother.rs
ParamSmoother::ParamSmoother(float val, uint32_t sample_rate, size_t max_block_frames, float smooth_secs,
                             float settle_eps) {
    // smoother.rs:93-112
    b = expf(-1.0f / (smooth_secs * (float)sample_rate));
    a = 1.0f - b;
    status = SmootherStatus::Inactive;
    input = val;
    output.assign(max_block_frames, val);
    last_output = val;
    settle_epsilon = settle_eps;
}

void ParamSmoother::reset(float val) {
    // smoother.rs:115-129
    if (is_active()) {
        status = SmootherStatus::Inactive;
        input = val;
        last_output = val;
        std::fill(output.begin(), output.end(), val);
    } else if (input != val) {
        input = val;
        last_output = val;
        std::fill(output.begin(), output.end(), val);
    }
}

void ParamSmoother::set(float val) {
    // smoother.rs:133-140
    if (input == val) return;
    input = val;
    status = SmootherStatus::Active;
}

SmoothedOutput ParamSmoother::process(size_t frames) {
    // smoother.rs:159-194
    frames = std::min(frames, output.size());
    if (status != SmootherStatus::Active || frames == 0 || output.empty()) {
        return SmoothedOutput{output.data(), output.size(), status};  // full-length slice (Q4)
    }
    float in = input * a;
    output[0] = in + (last_output * b);
    for (size_t i = 1; i < frames; ++i) output[i] = in + (output[i - 1] * b);
    last_output = output[frames - 1];
    switch (status) {
        case SmootherStatus::Active:
            if (fabsf(input - output[0]) < settle_epsilon) {  // Q1: tests output[0]
                reset(input);                                 // Q2: fills the whole buffer
                status = SmootherStatus::Deactivating;
            }
            break;
        case SmootherStatus::Deactivating:  // unreachable (Q3) but restated
            status = SmootherStatus::Inactive;
            break;
        default:
            break;
    }
    return SmoothedOutput{output.data(), frames, status};
}

// ================================================================= core/util.rs, core/param/range.rs
float db_to_gain(float db) { return powf(10.0f, 0.05f * db); }            // util.rs:7-9
float gain_to_db(float amp) { return 20.0f * log10f(amp); }               // util.rs:13-15
float db_to_gain_clamped_neg_100_db(float db) {                           // util.rs:21-27
    if (db <= -100.0f) return 0.0f;
    return db_to_gain(db);
}
float gain_to_db_clamped_neg_100_db(float amp) {  // util.rs:35-41
    if (amp <= 0.00001f) return -100.0f;
    return gain_to_db(amp);
}
float percent_volume_to_raw_gain(float percent_volume) {  // range.rs:32-35
    float n = fmaxf(percent_volume, 0.0f) * (1.0f / 100.0f);
    return n * n;
}

SilenceMask deinterleave(float* const* channels, size_t n_channels, size_t ch_len, const float* interleaved,
                         size_t interleaved_len, size_t num_interleaved_channels, bool calculate_silence_mask) {
    // util.rs:44-87.  `channels` is the iterator; each item has length ch_len.
    SilenceMask silence_mask;
    size_t i = 0;
    size_t next = 0;
    for (size_t k = 0; k < num_interleaved_channels; ++k) {
        if (next >= n_channels) return silence_mask;
        float* ch = channels[next++];
        if (calculate_silence_mask && i < 64) {
            // Q11: scans the DESTINATION channel's old contents.
            bool any = false;
            for (size_t f = 0; f < ch_len; ++f)
                if (ch[f] != 0.0f) {
                    any = true;
                    break;
                }
            if (!any) silence_mask.set_channel(i, true);
        }
        // interleaved.iter().skip(i).step_by(n).zip(ch.iter_mut())
        size_t f = 0;
        for (size_t src = i; src < interleaved_len && f < ch_len; src += num_interleaved_channels, ++f)
            ch[f] = interleaved[src];
        i += 1;
    }
    while (next < n_channels) {
        float* ch = channels[next++];
        for (size_t f = 0; f < ch_len; ++f) ch[f] = 0.0f;
        if (calculate_silence_mask && i < 64) silence_mask.set_channel(i, true);
        i += 1;
    }
    return silence_mask;
}

void interleave(const float* const* channels, size_t n_channels, size_t ch_len, float* interleaved,
                size_t interleaved_len, size_t num_interleaved_channels, const SilenceMask* mask) {
    // util.rs:90-120
    for (size_t k = 0; k < interleaved_len; ++k) interleaved[k] = 0.0f;
    size_t next = 0;
    for (size_t ch_i = 0; ch_i < num_interleaved_channels; ++ch_i) {
        if (next >= n_channels) return;
        const float* ch = channels[next++];
        if (mask && ch_i < 64 && mask->is_channel_silent(ch_i)) continue;
        size_t f = 0;
        for (size_t dst = ch_i; dst < interleaved_len && f < ch_len; dst += num_interleaved_channels, ++f)
            interleaved[dst] = ch[f];
    }
}

void interleave_stereo(const float* in_l, const float* in_r, float* interleaved, size_t interleaved_len,
                       const SilenceMask* mask) {
    // util.rs:123-147
    if (mask && mask->all_channels_silent(2)) {
        for (size_t k = 0; k < interleaved_len; ++k) interleaved[k] = 0.0f;
        return;
    }
    size_t frames = interleaved_len / 2;
    for (size_t f = 0; f < frames; ++f) {
        interleaved[2 * f] = in_l[f];
        interleaved[2 * f + 1] = in_r[f];
    }
}

void deinterleave_stereo(float* out_l, float* out_r, const float* interleaved, size_t interleaved_len) {
    // util.rs:150-162
    size_t frames = interleaved_len / 2;
    for (size_t f = 0; f < frames; ++f) {
        out_l[f] = interleaved[2 * f];
        out_r[f] = interleaved[2 * f + 1];
    }
}

void clear_all_outputs(size_t frames, float* const* outputs, size_t n_out, SilenceMask* out_mask) {
    // util.rs:165-175
    for (size_t c = 0; c < n_out; ++c)
        for (size_t f = 0; f < frames; ++f) outputs[c][f] = 0.0f;
    *out_mask = SilenceMask::new_all_silent(n_out);
}

// ================================================================= core/sample_resource.rs
float pcm_i16_to_f32(int16_t s) { return (float)s * (1.0f / 32767.0f); }             // :338-340
float pcm_u16_to_f32(uint16_t s) { return ((float)s * (2.0f / 65535.0f)) - 1.0f; }   // :343-345

namespace {
template <class T, class Conv>
void fill_buffers_interleaved(float* const* buffers, size_t n_buffers, size_t rs, size_t re,
                              uint64_t start_frame64, size_t channels, const T* data, Conv convert) {
    // sample_resource.rs:348-401
    size_t start_frame = (size_t)start_frame64;
    size_t frames = re - rs;
    if (channels == 1) {
        for (size_t i = 0; i < frames; ++i) buffers[0][rs + i] = convert(data[start_frame + i]);
        return;
    }
    if (channels == 2 && n_buffers >= 2) {
        const T* src = data + start_frame * 2;
        for (size_t i = 0; i < frames; ++i) {
            buffers[0][rs + i] = convert(src[2 * i]);
            buffers[1][rs + i] = convert(src[2 * i + 1]);
        }
        return;
    }
    const T* src = data + start_frame * channels;
    for (size_t ch = 0; ch < channels && ch < n_buffers; ++ch)
        for (size_t i = 0; i < frames; ++i) buffers[ch][rs + i] = convert(src[i * channels + ch]);
}

template <class T, class Conv>
void fill_buffers_deinterleaved(float* const* buffers, size_t n_buffers, size_t rs, size_t re,
                                uint64_t start_frame64, size_t channels, uint64_t len, const T* planes,
                                Conv convert) {
    // sample_resource.rs:404-439 (stereo fast path and generic path do the same arithmetic)
    size_t start_frame = (size_t)start_frame64;
    size_t frames = re - rs;
    for (size_t ch = 0; ch < channels && ch < n_buffers; ++ch) {
        const T* p = planes + ch * len + start_frame;
        for (size_t i = 0; i < frames; ++i) buffers[ch][rs + i] = convert(p[i]);
    }
}
}  // namespace

void SampleResource::fill_buffers(float* const* buffers, size_t n_buffers, size_t rs, size_t re,
                                  uint64_t start_frame) const {
    auto ident = [](float x) { return x; };
    switch (format) {
        case FMT_INTERLEAVED_I16:
            fill_buffers_interleaved(buffers, n_buffers, rs, re, start_frame, channels, i16.data(), pcm_i16_to_f32);
            break;
        case FMT_INTERLEAVED_U16:
            fill_buffers_interleaved(buffers, n_buffers, rs, re, start_frame, channels, u16.data(), pcm_u16_to_f32);
            break;
        case FMT_INTERLEAVED_F32:
            fill_buffers_interleaved(buffers, n_buffers, rs, re, start_frame, channels, f32.data(), ident);
            break;
        case FMT_PLANAR_I16:
            fill_buffers_deinterleaved(buffers, n_buffers, rs, re, start_frame, channels, frames, i16.data(),
                                       pcm_i16_to_f32);
            break;
        case FMT_PLANAR_U16:
            fill_buffers_deinterleaved(buffers, n_buffers, rs, re, start_frame, channels, frames, u16.data(),
                                       pcm_u16_to_f32);
            break;
        case FMT_PLANAR_F32:  // :442-456 copy_from_slice
            fill_buffers_deinterleaved(buffers, n_buffers, rs, re, start_frame, channels, frames, f32.data(), ident);
            break;
    }
}

// ================================================================= nodes/*.rs processors
namespace {

// ---- nodes/dummy.rs:33-42
struct DummyProcessor : AudioNodeProcessor {
    void process(size_t, const float* const*, size_t, float* const*, size_t, ProcInfo) override {}
};

// ---- nodes/volume.rs:79-145
struct VolumeProcessor : AudioNodeProcessor {
    std::shared_ptr<float> raw_gain;
    ParamSmoother gain_smoother;
    VolumeProcessor(std::shared_ptr<float> g, uint32_t sr, size_t mbf)
        : raw_gain(g), gain_smoother(*g, sr, mbf) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        float rg = *raw_gain;  // :92
        if (info.in_silence_mask.all_channels_silent(n_in)) {  // :94-100
            gain_smoother.reset(rg);
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        SmoothedOutput gain = gain_smoother.set_and_process(rg, frames);  // :102
        if (!gain.is_smoothing() && gain.values[0] < 0.00001f) {          // :104-108
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        *info.out_silence_mask = info.in_silence_mask;  // :110
        assert(frames <= gain.len);                     // :113
        if (n_in == 2 && n_out == 2) {                  // :116-129
            for (size_t i = 0; i < frames; ++i) {
                outputs[0][i] = inputs[0][i] * gain.values[i];
                outputs[1][i] = inputs[1][i] * gain.values[i];
            }
            return;
        }
        size_t n = std::min(n_in, n_out);  // zip
        for (size_t c = 0; c < n; ++c) {   // :131-143
            if (info.in_silence_mask.is_channel_silent(c)) {
                for (size_t i = 0; i < frames; ++i) outputs[c][i] = 0.0f;
                continue;
            }
            for (size_t i = 0; i < frames; ++i) outputs[c][i] = inputs[c][i] * gain.values[i];
        }
    }
};

// ---- nodes/sum.rs:37-136
struct SumNodeProcessor : AudioNodeProcessor {
    size_t num_in_ports;
    explicit SumNodeProcessor(size_t n) : num_in_ports(n) {}
    void process(size_t frames, const float* const* inputs, size_t num_inputs, float* const* outputs,
                 size_t num_outputs, ProcInfo info) override {
        if (info.in_silence_mask.all_channels_silent(num_inputs)) {  // :52-56
            clear_all_outputs(frames, outputs, num_outputs, info.out_silence_mask);
            return;
        }
        if (num_inputs == num_outputs) {  // :58-65 (Q14)
            for (size_t c = 0; c < num_outputs; ++c) memcpy(outputs[c], inputs[c], frames * sizeof(float));
            *info.out_silence_mask = info.in_silence_mask;
            return;
        }
        switch (num_in_ports) {
            case 2:  // :69-81
                for (size_t ch = 0; ch < num_outputs; ++ch) {
                    const float* in1 = inputs[ch];
                    const float* in2 = inputs[num_outputs * 1 + ch];
                    for (size_t i = 0; i < frames; ++i) outputs[ch][i] = in1[i] + in2[i];
                }
                break;
            case 3:  // :82-95
                for (size_t ch = 0; ch < num_outputs; ++ch) {
                    const float* in1 = inputs[ch];
                    const float* in2 = inputs[num_outputs * 1 + ch];
                    const float* in3 = inputs[num_outputs * 2 + ch];
                    for (size_t i = 0; i < frames; ++i) outputs[ch][i] = in1[i] + in2[i] + in3[i];
                }
                break;
            case 4:  // :96-110
                for (size_t ch = 0; ch < num_outputs; ++ch) {
                    const float* in1 = inputs[ch];
                    const float* in2 = inputs[num_outputs * 1 + ch];
                    const float* in3 = inputs[num_outputs * 2 + ch];
                    const float* in4 = inputs[num_outputs * 3 + ch];
                    for (size_t i = 0; i < frames; ++i) outputs[ch][i] = in1[i] + in2[i] + in3[i] + in4[i];
                }
                break;
            default: {  // :111-133 (Q13)
                size_t n = num_in_ports;
                for (size_t ch = 0; ch < num_outputs; ++ch) {
                    float* out = outputs[ch];
                    memcpy(out, inputs[ch], frames * sizeof(float));
                    for (size_t p = 1; p < n; ++p) {
                        size_t in_ch = num_outputs * p + ch;
                        if (info.in_silence_mask.is_channel_silent(in_ch)) continue;
                        const float* in = inputs[in_ch];
                        for (size_t i = 0; i < frames; ++i) out[i] += in[i];
                    }
                }
            }
        }
    }
};

// ---- nodes/sampler.rs:235-278
struct ProcLoopRange {
    uint64_t start, end;
    bool full_range;
};
static uint64_t sat_round_u64(double x) {
    // `(x).round() as u64`: round-half-away-from-zero then saturating cast (NaN -> 0)
    double r = round(x);
    if (!(r == r)) return 0;
    if (r <= 0.0) return 0;
    if (r >= 18446744073709551615.0) return ~0ull;
    return (uint64_t)r;
}

// ---- nodes/sampler.rs:280-561
struct SamplerProcessor : AudioNodeProcessor {
    std::shared_ptr<float> raw_gain;
    ParamSmoother gain_smoother;
    bool playing = false;
    uint32_t sample_rate;
    uint64_t playhead = 0;
    bool has_loop = false;
    ProcLoopRange loop_range{0, 0, false};
    std::shared_ptr<const SampleResource> sample;
    std::shared_ptr<std::deque<SamplerMsg>> from_node_rx;

    SamplerProcessor(std::shared_ptr<float> g, uint32_t sr, size_t mbf, std::shared_ptr<std::deque<SamplerMsg>> rx)
        : raw_gain(g), gain_smoother(*g, sr, mbf), sample_rate(sr), from_node_rx(rx) {}

    uint64_t loop_start_or_0() const { return has_loop ? loop_range.start : 0; }

    void process(size_t frames, const float* const*, size_t, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        // :331-414 drain messages
        while (!from_node_rx->empty()) {
            SamplerMsg msg = from_node_rx->front();
            from_node_rx->pop_front();
            switch (msg.type) {
                case SamplerMsg::SetSample:  // :333-364
                    sample = msg.sample;
                    if (has_loop && sample && loop_range.full_range) {  // update_sample :265-277
                        loop_range.start = 0;
                        loop_range.end = sample->len_frames();
                    }
                    if (msg.stop_playback) {
                        playhead = loop_start_or_0();
                        if (playing) playing = false;
                    }
                    break;
                case SamplerMsg::Play:  // :365-371
                    if (!playing) playing = true;
                    break;
                case SamplerMsg::Pause:  // :372-378
                    if (playing) playing = false;
                    break;
                case SamplerMsg::Stop:  // :379-391
                    playhead = loop_start_or_0();
                    if (playing) playing = false;
                    break;
                case SamplerMsg::SetPlayheadSecs: {  // :392-399
                    uint64_t s = sat_round_u64(msg.playhead_secs * (double)sample_rate);
                    if (s != playhead) playhead = s;
                    break;
                }
                case SamplerMsg::SetLoopRange:  // :400-412, ProcLoopRange::new :241-263
                    if (msg.loop_mode == 0) {
                        has_loop = false;
                    } else {
                        has_loop = true;
                        if (msg.loop_mode == 1) {
                            loop_range.start = 0;
                            loop_range.end = sample ? sample->len_frames() : 0;
                            loop_range.full_range = true;
                        } else {
                            loop_range.start = sat_round_u64(msg.loop_start * (double)sample_rate);
                            loop_range.end = sat_round_u64(msg.loop_end * (double)sample_rate);
                            loop_range.full_range = false;
                        }
                        // Q7: playhead INSIDE the new range snaps to its start
                        if (playhead >= loop_range.start && playhead < loop_range.end) playhead = loop_range.start;
                    }
                    break;
            }
        }
        if (!sample) {  // :416-422
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        if (!playing) {  // :424-430
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        float rg = *raw_gain;                                             // :432
        SmoothedOutput gain = gain_smoother.set_and_process(rg, frames);  // :433
        // :435 assert_eq!(gain.values.len(), frames)  (Q5).  The oracle's parity domain requires
        // frames == max_block_frames for sampler graphs whenever the smoother is not Active.
        assert(gain.len == frames && "Q5: reference would panic (frames != max_block_frames)");
        if (!gain.is_smoothing() && gain.values[0] < 0.00001f) {  // :437-443
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        if (has_loop) {  // :445-484
            if (playhead >= loop_range.end) playhead = loop_range.start;
            uint64_t left = loop_range.end - playhead;
            size_t frames_left = left <= (uint64_t)SIZE_MAX ? (size_t)left : SIZE_MAX;
            size_t first_copy_frames = std::min(frames, frames_left);
            sample->fill_buffers(outputs, n_out, 0, first_copy_frames, playhead);
            if (first_copy_frames < frames) {
                playhead = loop_range.start;
                size_t second_copy_frames = frames - first_copy_frames;
                sample->fill_buffers(outputs, n_out, first_copy_frames, frames, playhead);  // Q8: wraps once
                playhead += second_copy_frames;
            } else {
                playhead += frames;
            }
        } else {  // :485-517
            if (playhead >= sample->len_frames()) {
                playing = false;
                clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
                return;
            }
            size_t copy_frames = (size_t)std::min<uint64_t>(frames, sample->len_frames() - playhead);
            sample->fill_buffers(outputs, n_out, 0, copy_frames, playhead);
            if (copy_frames < frames) {  // Q9
                playing = false;
                playhead = 0;
                for (size_t c = 0; c < n_out; ++c)
                    for (size_t i = copy_frames; i < frames; ++i) outputs[c][i] = 0.0f;
            } else {
                playhead += frames;
            }
        }
        size_t sample_channels = sample->num_channels();  // :519
        if (n_out >= 2 && sample_channels == 2) {         // :523-533
            for (size_t i = 0; i < frames; ++i) {
                outputs[0][i] *= gain.values[i];
                outputs[1][i] *= gain.values[i];
            }
        } else {  // :535-542  zip(outputs, 0..sample_channels)
            size_t n = std::min(n_out, sample_channels);
            for (size_t c = 0; c < n; ++c)
                for (size_t i = 0; i < frames; ++i) outputs[c][i] *= gain.values[i];
        }
        if (n_out > sample_channels) {  // :545-559
            if (n_out == 2 && sample_channels == 1) {
                memcpy(outputs[1], outputs[0], frames * sizeof(float));
            } else {
                for (size_t c = sample_channels; c < n_out; ++c) {
                    for (size_t i = 0; i < frames; ++i) outputs[c][i] = 0.0f;
                    info.out_silence_mask->set_channel(c, true);
                }
            }
        }
    }
};

// ---- nodes/beep_test.rs:64-97
struct BeepTestProcessor : AudioNodeProcessor {
    std::shared_ptr<int> enabled;
    float phasor = 0.0f, phasor_inc, gain;
    BeepTestProcessor(std::shared_ptr<int> e, float inc, float g) : enabled(e), phasor_inc(inc), gain(g) {}
    void process(size_t frames, const float* const*, size_t, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        if (n_out == 0) return;  // :79-81
        float* out1 = outputs[0];
        if (!*enabled) {  // :83-86 (Q12: clears only outputs[1..], mask = new_all_silent(n-1))
            clear_all_outputs(frames, outputs + 1, n_out - 1, info.out_silence_mask);
            return;
        }
        const float TAU = 6.28318530717958647692528676655900577f;
        for (size_t i = 0; i < frames; ++i) {  // :88-91
            out1[i] = sinf(phasor * TAU) * gain;
            float t = phasor + phasor_inc;
            phasor = t - truncf(t);  // f32::fract
        }
        for (size_t c = 1; c < n_out; ++c) memcpy(outputs[c], out1, frames * sizeof(float));  // :93-95
    }
};

// ---- nodes/hard_clip.rs:47-95
struct HardClipProcessor : AudioNodeProcessor {
    float t;
    explicit HardClipProcessor(float th) : t(th) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        if (n_in == 2 && n_out == 2 && !info.in_silence_mask.any_channel_silent(2)) {  // :60-80 (Q16)
            for (size_t i = 0; i < frames; ++i) {
                outputs[0][i] = fmaxf(fminf(inputs[0][i], t), -t);
                outputs[1][i] = fmaxf(fminf(inputs[1][i], t), -t);
            }
            return;
        }
        size_t n = std::min(n_in, n_out);
        for (size_t c = 0; c < n; ++c) {  // :82-91
            if (info.in_silence_mask.is_channel_silent(c)) {
                for (size_t i = 0; i < frames; ++i) outputs[c][i] = 0.0f;
                continue;
            }
            // zip(output.iter_mut(), input.iter()): slices have length == frames (schedule.rs:373-377)
            for (size_t i = 0; i < frames; ++i) outputs[c][i] = fmaxf(fminf(inputs[c][i], t), -t);
        }
        *info.out_silence_mask = info.in_silence_mask;  // :93
    }
};

// ---- nodes/mono_to_stereo.rs:33-50
struct MonoToStereoProcessor : AudioNodeProcessor {
    void process(size_t frames, const float* const* inputs, size_t, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        if (info.in_silence_mask.is_channel_silent(0)) {
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        memcpy(outputs[0], inputs[0], frames * sizeof(float));
        memcpy(outputs[1], inputs[0], frames * sizeof(float));
    }
};

// ---- nodes/stereo_to_mono.rs:33-56
struct StereoToMonoProcessor : AudioNodeProcessor {
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        if (info.in_silence_mask.all_channels_silent(2) || n_in < 2 || n_out == 0) {
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        for (size_t i = 0; i < frames; ++i) outputs[0][i] = (inputs[0][i] + inputs[1][i]) * 0.5f;
    }
};

// ---- SPEC: stereo pan (NOT in the reference — DESIGN.md "spec nodes / pan").
// Control half computes the constant-power targets gl = cos(theta), gr = sin(theta),
// theta = (clamp(pan,-1,1)+1)*pi/4 in f64, rounded to f32 (exactly 1/0 at the ends).
// Audio half is VolumeProcessor's stereo path with one ParamSmoother per channel.
struct StereoPanProcessor : AudioNodeProcessor {
    std::shared_ptr<float> gl_t, gr_t;
    ParamSmoother sl, sr_;
    StereoPanProcessor(std::shared_ptr<float> l, std::shared_ptr<float> r, uint32_t sr, size_t mbf)
        : gl_t(l), gr_t(r), sl(*l, sr, mbf), sr_(*r, sr, mbf) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        float l = *gl_t, r = *gr_t;
        if (info.in_silence_mask.all_channels_silent(n_in)) {
            sl.reset(l);
            sr_.reset(r);
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        SmoothedOutput gl = sl.set_and_process(l, frames);
        SmoothedOutput gr = sr_.set_and_process(r, frames);
        *info.out_silence_mask = info.in_silence_mask;
        for (size_t i = 0; i < frames; ++i) {
            outputs[0][i] = inputs[0][i] * gl.values[i];
            outputs[1][i] = inputs[1][i] * gr.values[i];
        }
    }
};

// ---- SPEC: stereo width (mid/side), one smoothed parameter w >= 0 (1 = unchanged, 0 = mono)
struct StereoWidthProcessor : AudioNodeProcessor {
    std::shared_ptr<float> w_t;
    ParamSmoother sw;
    StereoWidthProcessor(std::shared_ptr<float> w, uint32_t sr, size_t mbf) : w_t(w), sw(*w, sr, mbf) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo info) override {
        float w = *w_t;
        if (info.in_silence_mask.all_channels_silent(n_in)) {
            sw.reset(w);
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        SmoothedOutput g = sw.set_and_process(w, frames);
        for (size_t i = 0; i < frames; ++i) {
            float l = inputs[0][i], r = inputs[1][i];
            float m = (l + r) * 0.5f;
            float sd = ((l - r) * 0.5f) * g.values[i];
            outputs[0][i] = m + sd;
            outputs[1][i] = m - sd;
        }
    }
};

// ---- SPEC: RBJ biquad, Direct Form I, f32 state.  y = b0 x + b1 x1 + b2 x2 - a2 y2 - a1 y1 evaluated as
//   ff = ((b0*x) + (b1*x1)) + (b2*x2)      feed-forward half: each product and sum rounded separately, left to right
//   y  = fma(-a1, y1, fma(-a2, y2, ff))    feedback half: two fused multiply-adds (Rust: f32::mul_add), older tap first
// so the recurrence's critical path is ONE fma per sample (y1 -> y); the feed-forward half does not depend on y and
// is computed ahead of it.  Coefficients are shared with the control half (5 floats).
struct BiquadProcessor : AudioNodeProcessor {
    std::shared_ptr<std::vector<float>> co;
    std::vector<float> st;  // [ch][x1 x2 y1 y2]
    BiquadProcessor(std::shared_ptr<std::vector<float>> c, size_t nch) : co(c), st(4 * nch, 0.0f) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo) override {
        size_t nch = std::min(n_in, n_out);
        const float b0 = (*co)[0], b1 = (*co)[1], b2 = (*co)[2], a1 = (*co)[3], a2 = (*co)[4];
        for (size_t c = 0; c < nch; ++c) {
            float x1 = st[4 * c], x2 = st[4 * c + 1], y1 = st[4 * c + 2], y2 = st[4 * c + 3];
            for (size_t i = 0; i < frames; ++i) {
                float x = inputs[c][i];
                float acc = b0 * x;
                acc = acc + (b1 * x1);
                acc = acc + (b2 * x2);
                acc = fmaf(-a2, y2, acc);
                acc = fmaf(-a1, y1, acc);
                x2 = x1;
                x1 = x;
                y2 = y1;
                y1 = acc;
                outputs[c][i] = acc;
            }
            st[4 * c] = x1;
            st[4 * c + 1] = x2;
            st[4 * c + 2] = y1;
            st[4 * c + 3] = y2;
        }
    }
};

// ---- SPEC: integer-sample delay with feedback: d = ring[pos]; ring[pos] = x + d*fb; out = x*dry + d*mix
struct DelayProcessor : AudioNodeProcessor {
    std::shared_ptr<float> fb, mix, dry;
    uint32_t D, pos = 0;
    std::vector<float> ring;  // [ch][D]
    DelayProcessor(std::shared_ptr<float> f, std::shared_ptr<float> m, std::shared_ptr<float> d, uint32_t D_, size_t nch)
        : fb(f), mix(m), dry(d), D(D_), ring((size_t)D_ * nch, 0.0f) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo) override {
        size_t nch = std::min(n_in, n_out);
        const float f = *fb, m = *mix, dr = *dry;
        for (size_t c = 0; c < nch; ++c) {
            float* r = ring.data() + c * D;
            uint32_t p = pos;
            for (size_t i = 0; i < frames; ++i) {
                float x = inputs[c][i];
                float d = r[p];
                r[p] = x + (d * f);
                outputs[c][i] = (x * dr) + (d * m);
                p = p + 1 == D ? 0 : p + 1;
            }
        }
        pos = (uint32_t)((pos + frames) % D);
    }
};

// ---- SPEC: FIR convolution y[n] = sum_k h[k] x[n-k] with a fully specified f32 summation order (the one a
// k-ordered fmaf chain per FIR_SEG-long window segment produces; DESIGN.md §6 "fir"):
//   window W = T-1+frames positions, position m holds x[n0-(T-1)+m]; H[m][i] = h[T-1-(m-i)] if 0<=m-i<=T-1 else 0
//   partial_s[i] = fmaf chain over m in segment s (ascending, from +0.0f), y[i] = (((p_0 + p_1) + p_2) + ...) + (+0.0f)
struct FirProcessor : AudioNodeProcessor {
    static constexpr size_t SEG = 4096;
    std::vector<std::vector<float>> h;     // per channel
    std::vector<std::vector<float>> hist;  // per channel: the last T-1 inputs
    size_t T;
    FirProcessor(const SampleResource& ir, size_t nch) {
        T = (size_t)ir.len_frames();
        for (size_t c = 0; c < nch; ++c) {
            size_t ic = std::min(c, ir.num_channels() - 1);
            std::vector<std::vector<float>> tmp(ir.num_channels(), std::vector<float>(T, 0.0f));
            std::vector<float*> ptrs;
            for (auto& v : tmp) ptrs.push_back(v.data());
            ir.fill_buffers(ptrs.data(), ptrs.size(), 0, T, 0);
            h.push_back(tmp[ic]);
            hist.emplace_back(T - 1, 0.0f);
        }
    }
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo) override {
        size_t nch = std::min(std::min(n_in, n_out), h.size());
        const size_t W = T - 1 + frames;
        std::vector<float> win(W);
        for (size_t c = 0; c < nch; ++c) {
            std::copy(hist[c].begin(), hist[c].end(), win.begin());
            std::copy(inputs[c], inputs[c] + frames, win.begin() + (T - 1));
            const float* hc = h[c].data();
            for (size_t i = 0; i < frames; ++i) {
                float total = 0.0f;
                for (size_t s0 = 0, sidx = 0; s0 < W; s0 += SEG, ++sidx) {
                    size_t s1 = std::min(W, s0 + SEG);
                    float acc = 0.0f;
                    for (size_t m = s0; m < s1; ++m) {
                        float hv = (m >= i && m - i <= T - 1) ? hc[T - 1 - (m - i)] : 0.0f;
                        acc = fmaf(win[m], hv, acc);
                    }
                    total = sidx == 0 ? acc : total + acc;
                }
                outputs[c][i] = total + 0.0f;  // SPEC: -0.0 (a sum that underflowed from below) is normalised to +0.0
            }
            if (T > 1) std::copy(win.end() - (T - 1), win.end(), hist[c].begin());
        }
    }
};

// ---- SPEC: resampling source (varispeed / sample-rate conversion): 0 inputs.  Source position is a 32.32
// fixed-point frame index advanced by `step` per output frame (exact integer arithmetic => bit-exact indexing);
// out[n] = sum_k h[phase][k] * s[idx - 7 + k], phase = top 5 fraction bits, as an ascending fmaf chain from +0.0.
// Outside [0, len) the source reads 0 (one-shot) or wraps (loop).  ctl = {step, playing, loop, seek_flag, seek_pos}
struct ResamplerProcessor : AudioNodeProcessor {
    std::shared_ptr<const SampleResource> src;
    std::shared_ptr<std::vector<double>> ctl;
    std::vector<float> h;
    uint64_t pos = 0;
    bool playing = false;
    ResamplerProcessor(std::shared_ptr<const SampleResource> s, std::shared_ptr<std::vector<double>> c)
        : src(s), ctl(c), h(RS_PHASES * RS_TAPS) {
        resampler_table(h.data());
    }
    float fetch(size_t c, int64_t j, bool loop) const {
        const int64_t len = (int64_t)src->len_frames();
        if (loop) {
            j %= len;
            if (j < 0) j += len;
        } else if (j < 0 || j >= len) {
            return 0.0f;
        }
        float v = 0.0f;
        float* bufs[64];
        float tmp[64];
        for (size_t k = 0; k < src->num_channels() && k < 64; ++k) bufs[k] = &tmp[k];
        src->fill_buffers(bufs, std::min<size_t>(src->num_channels(), 64), 0, 1, (uint64_t)j);
        v = tmp[c];
        return v;
    }
    void process(size_t frames, const float* const*, size_t, float* const* outputs, size_t n_out, ProcInfo info) override {
        std::vector<double>& cw = *ctl;
        const uint64_t step = (uint64_t)cw[0];
        const bool loop = cw[2] != 0.0;
        if (cw[3] != 0.0) {  // seek message
            pos = ((uint64_t)cw[4]) << 32;
            cw[3] = 0.0;
        }
        playing = cw[1] != 0.0;
        const uint64_t len = src ? src->len_frames() : 0;
        if (!playing || len == 0) {
            clear_all_outputs(frames, outputs, n_out, info.out_silence_mask);
            return;
        }
        const size_t sch = src->num_channels();
        const size_t nfill = std::min(n_out, sch);
        for (size_t i = 0; i < frames; ++i) {
            const uint64_t p = pos + (uint64_t)i * step;
            const int64_t idx = (int64_t)(p >> 32);
            const uint32_t ph = (uint32_t)(p >> 27) & (RS_PHASES - 1);
            const float* hp = h.data() + ph * RS_TAPS;
            for (size_t c = 0; c < nfill; ++c) {
                float acc = 0.0f;
                for (int k = 0; k < RS_TAPS; ++k) acc = fmaf(hp[k], fetch(c, idx - (RS_TAPS / 2 - 1) + k, loop), acc);
                outputs[c][i] = acc;
            }
        }
        if (n_out > sch) {  // like the sampler (sampler.rs:545-559): mono -> both outputs, else zero + flag
            if (n_out == 2 && sch == 1) {
                for (size_t i = 0; i < frames; ++i) outputs[1][i] = outputs[0][i];
            } else {
                for (size_t c = sch; c < n_out; ++c) {
                    for (size_t i = 0; i < frames; ++i) outputs[c][i] = 0.0f;
                    info.out_silence_mask->set_channel(c, true);
                }
            }
        }
        pos += (uint64_t)frames * step;
        if (loop) {
            pos %= (len << 32);
        } else if ((pos >> 32) >= len + RS_TAPS / 2) {
            cw[1] = 0.0;  // ran off the end: stops before the next block
        }
    }
};

// ---- SPEC: 3D spatialiser (listener at the origin): inverse-distance attenuation, equal-power pan from the
// direction cosine to the right, per-ear integer delay (ITD).  Mono sum m = in0 or (in0+in1)*0.5; history of the
// last SP_HIST mono samples; outL[i] = M(i-dl)*gL[i], outR[i] = M(i-dr)*gR[i], gains through ParamSmoothers.
struct SpatialProcessor : AudioNodeProcessor {
    std::shared_ptr<float> gl_t, gr_t;
    std::shared_ptr<std::vector<double>> ctl;  // {dl, dr}
    ParamSmoother sl, sr_;
    std::vector<float> hist;
    SpatialProcessor(std::shared_ptr<float> l, std::shared_ptr<float> r, std::shared_ptr<std::vector<double>> c, uint32_t sr,
                     size_t mbf)
        : gl_t(l), gr_t(r), ctl(c), sl(*l, sr, mbf), sr_(*r, sr, mbf), hist(SP_HIST, 0.0f) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out,
                 ProcInfo) override {
        const int dl = (int)(*ctl)[0], dr = (int)(*ctl)[1];
        SmoothedOutput gl = sl.set_and_process(*gl_t, frames);
        SmoothedOutput gr = sr_.set_and_process(*gr_t, frames);
        std::vector<float> m(frames);
        for (size_t i = 0; i < frames; ++i) m[i] = n_in >= 2 ? (inputs[0][i] + inputs[1][i]) * 0.5f : inputs[0][i];
        auto M = [&](int64_t j) -> float { return j >= 0 ? m[(size_t)j] : hist[(size_t)(SP_HIST + j)]; };
        for (size_t i = 0; i < frames; ++i) {
            outputs[0][i] = M((int64_t)i - dl) * gl.values[i];
            if (n_out > 1) outputs[1][i] = M((int64_t)i - dr) * gr.values[i];
        }
        std::vector<float> nh(SP_HIST);
        for (int l = 0; l < SP_HIST; ++l) {
            int64_t j = (int64_t)frames - SP_HIST + l;  // index into hist ++ m, relative to m[0]
            nh[l] = M(j);
        }
        hist.swap(nh);
    }
};

}  // namespace

// Kaiser-windowed sinc, cutoff 0.9 x Nyquist, beta 8; each phase normalised to unity DC gain in f64, then f32.
static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 64; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
    }
    return sum;
}
void resampler_table(float* h) {
    const double fc = 0.9, beta = 8.0, half = RS_TAPS / 2.0, pi = 3.14159265358979323846;
    const double i0b = bessel_i0(beta);
    for (int ph = 0; ph < RS_PHASES; ++ph) {
        double row[RS_TAPS], sum = 0.0;
        for (int k = 0; k < RS_TAPS; ++k) {
            double t = (double)(k - (RS_TAPS / 2 - 1)) - (double)ph / RS_PHASES;  // tap position relative to the sample point
            double x = pi * fc * t;
            double sinc = fabs(t) < 1e-12 ? 1.0 : sin(x) / x;
            double r = t / half;
            double w = fabs(r) >= 1.0 ? 0.0 : bessel_i0(beta * sqrt(1.0 - r * r)) / i0b;
            row[k] = fc * sinc * w;
            sum += row[k];
        }
        for (int k = 0; k < RS_TAPS; ++k) h[ph * RS_TAPS + k] = (float)(row[k] / sum);
    }
}
uint64_t resampler_step(float ratio) {
    double r = (double)ratio;
    if (!(r >= 1.0 / 256.0)) r = 1.0 / 256.0;
    if (r > 256.0) r = 256.0;
    return (uint64_t)llround(r * 4294967296.0);
}
void spatial_params(float x, float y, float z, uint32_t sample_rate, float* gl, float* gr, int* dl, int* dr) {
    const double pi = 3.14159265358979323846;
    double d = sqrt((double)x * x + (double)y * y + (double)z * z);
    double att = 1.0 / fmax(d, 1.0);                 // inverse distance, reference distance 1, rolloff 1
    double s = d < 1e-9 ? 0.0 : (double)x / d;       // direction cosine to the right, in [-1, 1]
    double theta = (s + 1.0) * (pi / 4.0);
    *gl = (float)(cos(theta) * att);
    *gr = (float)(sin(theta) * att);
    double itd_max = round(0.00066 * (double)sample_rate);
    if (itd_max > SP_HIST - 1) itd_max = SP_HIST - 1;
    *dl = (int)round(fmax(0.0, s) * itd_max);        // source on the right: the left ear hears it later
    *dr = (int)round(fmax(0.0, -s) * itd_max);
}

void biquad_coefs(int type, float cutoff_hz, float q, uint32_t sample_rate, float co[5]) {
    double fs = (double)sample_rate;
    double f0 = fmin(fmax((double)cutoff_hz, 1.0), 0.49 * fs);
    double Q = fmax((double)q, 1e-3);
    double w0 = 2.0 * 3.14159265358979323846 * f0 / fs;
    double cw = cos(w0), alpha = sin(w0) / (2.0 * Q);
    double b0, b1, b2, a0 = 1.0 + alpha, a1 = -2.0 * cw, a2 = 1.0 - alpha;
    if (type == 1) {
        b0 = (1.0 + cw) * 0.5;
        b1 = -(1.0 + cw);
        b2 = (1.0 + cw) * 0.5;
    } else if (type == 2) {
        b0 = alpha;
        b1 = 0.0;
        b2 = -alpha;
    } else {
        b0 = (1.0 - cw) * 0.5;
        b1 = 1.0 - cw;
        b2 = (1.0 - cw) * 0.5;
    }
    co[0] = (float)(b0 / a0);
    co[1] = (float)(b1 / a0);
    co[2] = (float)(b2 / a0);
    co[3] = (float)(a1 / a0);
    co[4] = (float)(a2 / a0);
}

void pan_to_gains(float pan, float* gl, float* gr) {
    float p = fminf(fmaxf(pan, -1.0f), 1.0f);
    if (p <= -1.0f) {
        *gl = 1.0f;
        *gr = 0.0f;
        return;
    }
    if (p >= 1.0f) {
        *gl = 0.0f;
        *gr = 1.0f;
        return;
    }
    double theta = ((double)p + 1.0) * (3.14159265358979323846 / 4.0);
    *gl = (float)cos(theta);
    *gr = (float)sin(theta);
}

// ================================================================= control halves
const char* AudioNode::debug_name() const {
    switch (kind) {
        case KIND_DUMMY: return "dummy";
        case KIND_BEEP_TEST: return "beep_test";
        case KIND_VOLUME: return "volume";
        case KIND_SUM: return "sum";
        case KIND_SAMPLER: return "beep_test";  // Q23 (sampler.rs:185-187)
        case KIND_HARD_CLIP: return "hard_clip";
        case KIND_MONO_TO_STEREO: return "mono_to_stereo";
        case KIND_STEREO_TO_MONO: return "stereo_to_mono";
        case KIND_STEREO_PAN: return "stereo_pan";
        case KIND_STEREO_WIDTH: return "stereo_width";
        case KIND_BIQUAD: return "biquad";
        case KIND_DELAY: return "delay";
        case KIND_FIR: return "fir";
        case KIND_RESAMPLER: return "resampler";
        case KIND_SPATIAL: return "spatial";
        default: return "unknown";
    }
}

std::unique_ptr<AudioNode> make_node(int kind, const float* params, int n_params) {
    auto p = [&](int i, float d) { return i < n_params ? params[i] : d; };
    std::unique_ptr<AudioNode> n(new AudioNode());
    n->kind = kind;
    switch (kind) {
        case KIND_VOLUME:   // volume.rs:15-22
        case KIND_SAMPLER:  // sampler.rs:55-64
            n->raw_gain = std::make_shared<float>(percent_volume_to_raw_gain(fmaxf(p(0, 100.0f), 0.0f)));
            if (kind == KIND_SAMPLER) n->to_processor = std::make_shared<std::deque<SamplerMsg>>();
            break;
        case KIND_BEEP_TEST: {  // beep_test.rs:15-24
            float f = p(0, 440.0f);
            // f32::clamp(20, 20000): NaN stays NaN
            if (f < 20.0f) f = 20.0f;
            if (f > 20000.0f) f = 20000.0f;
            n->freq_hz = f;
            float g = db_to_gain_clamped_neg_100_db(p(1, -12.0f));
            if (g < 0.0f) g = 0.0f;
            if (g > 1.0f) g = 1.0f;
            n->gain = g;
            n->enabled = std::make_shared<int>(p(2, 1.0f) != 0.0f ? 1 : 0);
            break;
        }
        case KIND_HARD_CLIP:  // hard_clip.rs:8-12
            n->threshold_gain = db_to_gain_clamped_neg_100_db(p(0, 0.0f));
            break;
        case KIND_STEREO_PAN: {
            float gl, gr;
            pan_to_gains(p(0, 0.0f), &gl, &gr);
            n->aux0 = std::make_shared<float>(gl);
            n->aux1 = std::make_shared<float>(gr);
            break;
        }
        case KIND_STEREO_WIDTH:
            n->aux0 = std::make_shared<float>(fmaxf(p(0, 1.0f), 0.0f));
            break;
        case KIND_BIQUAD:
            n->spec_params = {p(0, 0.0f), p(1, 1000.0f), p(2, 0.70710678f)};
            n->coefs = std::make_shared<std::vector<float>>(5, 0.0f);
            break;
        case KIND_DELAY: {
            float mix = fminf(fmaxf(p(2, 0.5f), 0.0f), 1.0f);
            n->spec_params = {p(0, 0.1f)};
            n->raw_gain = std::make_shared<float>(fminf(fmaxf(p(1, 0.0f), 0.0f), 0.999f));  // feedback
            n->aux0 = std::make_shared<float>(mix);
            n->aux1 = std::make_shared<float>(1.0f - mix);
            break;
        }
        case KIND_RESAMPLER:  // params: sample id, ratio, loop, playing
            n->ctl = std::make_shared<std::vector<double>>(
                std::vector<double>{(double)resampler_step(p(1, 1.0f)), p(3, 1.0f) != 0.0f ? 1.0 : 0.0,
                                    p(2, 0.0f) != 0.0f ? 1.0 : 0.0, 0.0, 0.0});
            break;
        case KIND_SPATIAL:  // params: x, y, z
            n->spec_params = {p(0, 0.0f), p(1, 0.0f), p(2, -1.0f)};
            n->aux0 = std::make_shared<float>(0.0f);
            n->aux1 = std::make_shared<float>(0.0f);
            n->ctl = std::make_shared<std::vector<double>>(std::vector<double>{0.0, 0.0});
            break;
        default:
            break;
    }
    return n;
}

// KIND_CUSTOM: a node the graph knows nothing about — processor.rs:243 calls it like any other
struct CustomProcessor : AudioNodeProcessor {
    AudioNode::CustomFn fn;
    void* user;
    CustomProcessor(AudioNode::CustomFn f, void* u) : fn(f), user(u) {}
    void process(size_t frames, const float* const* inputs, size_t n_in, float* const* outputs, size_t n_out, ProcInfo info) override {
        uint64_t om = info.out_silence_mask->bits;
        fn(user, frames, inputs, (uint32_t)n_in, outputs, (uint32_t)n_out, info.in_silence_mask.bits, &om, info.stream_time_secs, info.stream_status);
        info.out_silence_mask->bits = om;
    }
};

std::unique_ptr<AudioNodeProcessor> AudioNode::activate(uint32_t sample_rate, size_t max_block_frames,
                                                        size_t num_inputs, size_t num_outputs, std::string& err) {
    switch (kind) {
        case KIND_CUSTOM:
            if (!custom_fn) {
                err = "custom node without a process function";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(new CustomProcessor(custom_fn, custom_user));
        case KIND_DUMMY:
            return std::unique_ptr<AudioNodeProcessor>(new DummyProcessor());
        case KIND_VOLUME:  // volume.rs:56-76
            if (num_inputs != num_outputs) {
                err = "The number of inputs on a VolumeNode node must equal the number of outputs.";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(new VolumeProcessor(raw_gain, sample_rate, max_block_frames));
        case KIND_SUM:  // sum.rs:20-34
            if (num_outputs == 0 || num_inputs % num_outputs != 0) {
                err = "The number of inputs on a SumNode must be a multiple of the number of outputs.";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(new SumNodeProcessor(num_inputs / num_outputs));
        case KIND_SAMPLER:  // sampler.rs:198-222
            return std::unique_ptr<AudioNodeProcessor>(
                new SamplerProcessor(raw_gain, sample_rate, max_block_frames, to_processor));
        case KIND_BEEP_TEST:  // beep_test.rs:48-61
            return std::unique_ptr<AudioNodeProcessor>(
                new BeepTestProcessor(enabled, freq_hz / (float)sample_rate, gain));
        case KIND_HARD_CLIP:  // hard_clip.rs:30-44
            if (num_inputs != num_outputs) {
                err = "The number of inputs on a HardClip node must equal the number of outputs.";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(new HardClipProcessor(threshold_gain));
        case KIND_MONO_TO_STEREO:
            return std::unique_ptr<AudioNodeProcessor>(new MonoToStereoProcessor());
        case KIND_STEREO_TO_MONO:
            return std::unique_ptr<AudioNodeProcessor>(new StereoToMonoProcessor());
        case KIND_STEREO_PAN:
            if (num_inputs != 2 || num_outputs != 2) {
                err = "StereoPanNode needs exactly 2 inputs and 2 outputs.";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(
                new StereoPanProcessor(aux0, aux1, sample_rate, max_block_frames));
        case KIND_STEREO_WIDTH:
            if (num_inputs != 2 || num_outputs != 2) {
                err = "StereoWidthNode needs exactly 2 inputs and 2 outputs.";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(new StereoWidthProcessor(aux0, sample_rate, max_block_frames));
        case KIND_BIQUAD: {
            if (num_inputs != num_outputs || num_inputs == 0) {
                err = "Biquad/Delay nodes need as many outputs as inputs (>= 1).";
                return nullptr;
            }
            biquad_coefs((int)spec_params[0], spec_params[1], spec_params[2], sample_rate, coefs->data());
            act_sample_rate = sample_rate;
            return std::unique_ptr<AudioNodeProcessor>(new BiquadProcessor(coefs, num_inputs));
        }
        case KIND_DELAY: {
            if (num_inputs != num_outputs || num_inputs == 0) {
                err = "Biquad/Delay nodes need as many outputs as inputs (>= 1).";
                return nullptr;
            }
            double d = round((double)spec_params[0] * (double)sample_rate);
            if (!(d >= 1.0)) d = 1.0;
            if (d > 16777216.0) d = 16777216.0;
            return std::unique_ptr<AudioNodeProcessor>(new DelayProcessor(raw_gain, aux0, aux1, (uint32_t)d, num_inputs));
        }
        case KIND_FIR:
            if (num_inputs != num_outputs || num_inputs == 0 || !ir || ir->len_frames() == 0) {
                err = "FIR node needs as many outputs as inputs and a non-empty impulse response.";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(new FirProcessor(*ir, num_inputs));
        case KIND_RESAMPLER:
            if (num_inputs != 0 || num_outputs == 0 || !ir) {
                err = "Resampler node is a source: 0 inputs, >= 1 outputs and a source sample.";
                return nullptr;
            }
            return std::unique_ptr<AudioNodeProcessor>(new ResamplerProcessor(ir, ctl));
        case KIND_SPATIAL: {
            if (!(num_inputs == 1 || num_inputs == 2) || num_outputs != 2) {
                err = "Spatial node needs 1 or 2 inputs and exactly 2 outputs.";
                return nullptr;
            }
            act_sample_rate = sample_rate;
            int dl, dr;
            spatial_params(spec_params[0], spec_params[1], spec_params[2], sample_rate, aux0.get(), aux1.get(), &dl, &dr);
            (*ctl)[0] = dl;
            (*ctl)[1] = dr;
            return std::unique_ptr<AudioNodeProcessor>(new SpatialProcessor(aux0, aux1, ctl, sample_rate, max_block_frames));
        }
        default:
            err = "unknown node kind";
            return nullptr;
    }
}

// ================================================================= graph/graph/compiler/schedule.rs
CompiledSchedule::CompiledSchedule(std::vector<ScheduledNode> s, size_t nb, size_t mbf)
    : schedule(std::move(s)), buffers(nb * mbf, 0.0f), buffer_silence_flags(nb, 0), num_buffers(nb),
      max_block_frames(mbf) {}  // schedule.rs:195-207

void CompiledSchedule::prepare_graph_inputs(size_t frames, size_t num_stream_inputs,
                                            const std::function<SilenceMask(float* const*, size_t)>& fill_inputs) {
    // schedule.rs:213-253
    frames = std::min(frames, max_block_frames);
    ScheduledNode& graph_in_node = schedule.front();
    std::vector<float*> inputs;
    size_t fill_input_len = std::min(num_stream_inputs, graph_in_node.output_buffers.size());
    for (size_t i = 0; i < fill_input_len; ++i)
        inputs.push_back(buffer_slice(graph_in_node.output_buffers[i].buffer_index));
    SilenceMask silence_mask = fill_inputs(inputs.data(), inputs.size());
    for (size_t i = 0; i < fill_input_len; ++i)
        buffer_silence_flags[graph_in_node.output_buffers[i].buffer_index] = silence_mask.is_channel_silent(i);
    if (fill_input_len < graph_in_node.output_buffers.size()) {
        for (size_t k = fill_input_len; k < graph_in_node.output_buffers.size(); ++k) {
            float* b = buffer_slice(graph_in_node.output_buffers[k].buffer_index);
            for (size_t f = 0; f < frames; ++f) b[f] = 0.0f;
            buffer_silence_flags[graph_in_node.output_buffers[k].buffer_index] = 1;
        }
    }
}

void CompiledSchedule::read_graph_outputs(
    size_t frames, size_t num_stream_outputs,
    const std::function<void(const float* const*, size_t, SilenceMask)>& read_outputs) {
    // schedule.rs:255-287
    frames = std::min(frames, max_block_frames);
    (void)frames;
    ScheduledNode& graph_out_node = schedule.back();
    std::vector<const float*> outputs;
    SilenceMask silence_mask;
    size_t read_output_len = std::min(num_stream_outputs, graph_out_node.input_buffers.size());
    for (size_t i = 0; i < read_output_len; ++i) {
        size_t bi = graph_out_node.input_buffers[i].buffer_index;
        if (buffer_silence_flags[bi]) silence_mask.set_channel(i, true);
        outputs.push_back(buffer_slice(bi));
    }
    read_outputs(outputs.data(), outputs.size(), silence_mask);
}

void CompiledSchedule::process(size_t frames,
                               const std::function<SilenceMask(NodeID, SilenceMask, const float* const*, size_t,
                                                               float* const*, size_t)>& process) {
    // schedule.rs:289-344
    frames = std::min(frames, max_block_frames);
    std::vector<const float*> inputs;
    std::vector<float*> outputs;
    for (ScheduledNode& sn : schedule) {
        SilenceMask in_silence_mask;
        inputs.clear();
        outputs.clear();
        for (size_t i = 0; i < sn.input_buffers.size(); ++i) {
            const InBufferAssignment& b = sn.input_buffers[i];
            float* buf = buffer_slice(b.buffer_index);
            if (b.should_clear) {
                for (size_t f = 0; f < frames; ++f) buf[f] = 0.0f;
                buffer_silence_flags[b.buffer_index] = 1;
            }
            if (buffer_silence_flags[b.buffer_index]) in_silence_mask.set_channel(i, true);
            inputs.push_back(buf);
        }
        for (const OutBufferAssignment& b : sn.output_buffers) outputs.push_back(buffer_slice(b.buffer_index));
        SilenceMask out_silence_mask =
            process(sn.id, in_silence_mask, inputs.data(), inputs.size(), outputs.data(), outputs.size());
        for (size_t i = 0; i < sn.output_buffers.size(); ++i)
            buffer_silence_flags[sn.output_buffers[i].buffer_index] = out_silence_mask.is_channel_silent(i);
    }
}

// ================================================================= graph/graph/compiler.rs
namespace {
struct BufferRef {  // compiler.rs:82-88
    size_t idx;
    size_t generation;
};
struct BufferAllocator {  // compiler.rs:92-136
    std::vector<BufferRef> free_list;
    size_t count = 0;
    std::shared_ptr<BufferRef> acquire() {
        BufferRef entry;
        if (!free_list.empty()) {
            entry = free_list.back();  // Vec::pop — LIFO
            free_list.pop_back();
        } else {
            entry.idx = count;
            count += 1;
            entry.generation = 0;
        }
        return std::make_shared<BufferRef>(entry);
    }
    void release(std::shared_ptr<BufferRef>& r) {
        if (r.use_count() == 1) free_list.push_back(BufferRef{r->idx, r->generation + 1});
    }
};

void preprocess(Arena<NodeEntry>& nodes, Arena<Edge>& edges) {
    // compiler.rs:191-228
    nodes.for_each([&](Index, NodeEntry& n) {
        assert(n.num_inputs <= 64 && n.num_outputs <= 64);
        n.incoming.clear();
        n.outgoing.clear();
    });
    edges.for_each([&](Index, Edge& e) {
        nodes.get(e.src_node)->outgoing.push_back(e);
        nodes.get(e.dst_node)->incoming.push_back(e);
    });
}

// compiler.rs:232-300.  Returns false when a cycle is detected.
bool sort_topologically(Arena<NodeEntry>& nodes, NodeID graph_in, NodeID graph_out, bool build_schedule,
                        std::vector<ScheduledNode>& schedule) {
    std::vector<int32_t> in_degree(nodes.capacity(), 0);
    std::deque<uint32_t> queue;
    size_t num_visited = 0;
    nodes.for_each([&](Index, NodeEntry& n) {
        for (const Edge& e : n.outgoing) in_degree[e.dst_node.slot] += 1;
    });
    queue.push_back(graph_in.slot);
    nodes.for_each([&](Index, NodeEntry& n) {
        if (n.incoming.empty() && n.id.slot != graph_in.slot) queue.push_back(n.id.slot);
    });
    while (!queue.empty()) {
        uint32_t slot = queue.front();
        queue.pop_front();
        num_visited += 1;
        NodeEntry* n = nodes.get_by_slot(slot);
        for (const Edge& e : n->outgoing) {
            in_degree[e.dst_node.slot] -= 1;
            if (in_degree[e.dst_node.slot] == 0) queue.push_back(e.dst_node.slot);
        }
        if (build_schedule && slot != graph_out.slot) {
            ScheduledNode sn;
            sn.id = n->id;
            schedule.push_back(sn);
        }
    }
    if (build_schedule) {
        ScheduledNode sn;
        sn.id = graph_out;
        schedule.push_back(sn);
    }
    return num_visited == nodes.len;
}
}  // namespace

int compile(Arena<NodeEntry>& nodes, Arena<Edge>& edges, NodeID graph_in, NodeID graph_out,
            size_t max_block_frames, std::unique_ptr<CompiledSchedule>& out) {
    // compiler.rs:139-152
    preprocess(nodes, edges);
    std::vector<ScheduledNode> schedule;
    if (!sort_topologically(nodes, graph_in, graph_out, true, schedule)) return ERR_COMPILE_CYCLE;

    // solve_buffer_requirements, compiler.rs:302-412
    BufferAllocator allocator;
    std::map<int64_t, std::shared_ptr<BufferRef>> assignment_table;  // keyed by edge id
    std::vector<std::shared_ptr<BufferRef>> buffers_to_release;
    for (ScheduledNode& entry : schedule) {
        NodeEntry* ne = nodes.get(entry.id);
        buffers_to_release.clear();
        for (uint32_t port = 0; port < ne->num_inputs; ++port) {
            std::vector<const Edge*> es;
            for (const Edge& e : ne->incoming)
                if (e.dst_port == port) es.push_back(&e);
            if (es.empty()) {
                auto buffer = allocator.acquire();
                entry.input_buffers.push_back(InBufferAssignment{buffer->idx, true, buffer->generation});
                buffers_to_release.push_back(buffer);
            } else if (es.size() == 1) {
                auto it = assignment_table.find(index_to_i64(es[0]->id));
                assert(it != assignment_table.end() && "No buffer assigned to edge!");
                auto buffer = it->second;
                assignment_table.erase(it);
                entry.input_buffers.push_back(InBufferAssignment{buffer->idx, false, buffer->generation});
                buffers_to_release.push_back(buffer);
            } else {
                return ERR_COMPILE_MANY_TO_ONE;
            }
        }
        for (uint32_t port = 0; port < ne->num_outputs; ++port) {
            std::vector<const Edge*> es;
            for (const Edge& e : ne->outgoing)
                if (e.src_port == port) es.push_back(&e);
            if (es.empty()) {
                auto buffer = allocator.acquire();
                entry.output_buffers.push_back(OutBufferAssignment{buffer->idx, buffer->generation});
                buffers_to_release.push_back(buffer);
            } else {
                auto buffer = allocator.acquire();
                for (const Edge* e : es) assignment_table[index_to_i64(e->id)] = buffer;
                entry.output_buffers.push_back(OutBufferAssignment{buffer->idx, buffer->generation});
            }
        }
        // drain(..): each element is released while later elements are still alive in the Vec
        for (size_t k = 0; k < buffers_to_release.size(); ++k) {
            std::shared_ptr<BufferRef> b = std::move(buffers_to_release[k]);
            allocator.release(b);
        }
        buffers_to_release.clear();
    }
    out.reset(new CompiledSchedule(std::move(schedule), allocator.count, max_block_frames));  // merge :415-417
    return 0;
}

bool cycle_detected(Arena<NodeEntry>& nodes, Arena<Edge>& edges, NodeID graph_in, NodeID graph_out) {
    // compiler.rs:154-168
    preprocess(nodes, edges);
    std::vector<ScheduledNode> unused;
    return !sort_topologically(nodes, graph_in, graph_out, false, unused);
}

// ================================================================= graph/graph.rs
AudioGraph::AudioGraph(size_t num_graph_inputs, size_t num_graph_outputs) {
    // graph.rs:125-168
    {
        std::unique_ptr<NodeEntry> e(new NodeEntry());
        e->num_inputs = 0;
        e->num_outputs = (uint32_t)num_graph_inputs;
        e->node = make_node(KIND_DUMMY, nullptr, 0);
        graph_in_id = nodes.insert(std::move(e));
        nodes.get(graph_in_id)->id = graph_in_id;
    }
    {
        std::unique_ptr<NodeEntry> e(new NodeEntry());
        e->num_inputs = (uint32_t)num_graph_outputs;
        e->num_outputs = 0;
        e->node = make_node(KIND_DUMMY, nullptr, 0);
        graph_out_id = nodes.insert(std::move(e));
        nodes.get(graph_out_id)->id = graph_out_id;
    }
    needs_compile = true;
    nodes_to_activate = {graph_in_id, graph_out_id};
}

NodeID AudioGraph::add_node(size_t num_inputs, size_t num_outputs, std::unique_ptr<AudioNode> node) {
    // graph.rs:201-231 (Q25: no validation against AudioNodeInfo)
    std::unique_ptr<NodeEntry> e(new NodeEntry());
    e->num_inputs = (uint32_t)num_inputs;
    e->num_outputs = (uint32_t)num_outputs;
    e->node = std::move(node);
    NodeID id = nodes.insert(std::move(e));
    nodes.get(id)->id = id;
    nodes_to_activate.push_back(id);
    needs_compile = true;
    return id;
}

int AudioGraph::remove_node(NodeID id) {
    // graph.rs:268-299
    if (id == graph_in_id || id == graph_out_id) return -1;
    std::unique_ptr<NodeEntry> e = nodes.remove(id);
    if (!e) return -1;
    for (uint32_t p = 0; p < e->num_inputs; ++p) remove_edges_with_input_port(id, p);
    for (uint32_t p = 0; p < e->num_outputs; ++p) remove_edges_with_output_port(id, p);
    for (uint32_t p = 0; p < e->num_inputs; ++p) connected_input_ports.erase({index_to_i64(id), p});
    nodes_to_remove_from_schedule.push_back(id);
    needs_compile = true;
    return 0;
}

int64_t AudioGraph::connect(NodeID src, uint32_t src_port, NodeID dst, uint32_t dst_port, bool check_for_cycles) {
    // graph.rs:396-477
    NodeEntry* s = nodes.get(src);
    if (!s) return ERR_SRC_NODE_NOT_FOUND;
    NodeEntry* d = nodes.get(dst);
    if (!d) return ERR_DST_NODE_NOT_FOUND;
    if (src_port >= s->num_outputs) return ERR_OUT_PORT_OUT_OF_RANGE;
    if (dst_port >= d->num_inputs) return ERR_IN_PORT_OUT_OF_RANGE;
    if (src == dst) return ERR_CYCLE_DETECTED;
    auto key = std::make_tuple(index_to_i64(src), src_port, index_to_i64(dst), dst_port);
    if (existing_edges.count(key)) return ERR_EDGE_ALREADY_EXISTS;
    auto ipk = std::make_pair(index_to_i64(dst), dst_port);
    if (connected_input_ports.count(ipk)) return ERR_INPUT_PORT_ALREADY_CONNECTED;
    connected_input_ports[ipk] = true;
    std::unique_ptr<Edge> e(new Edge());
    e->src_node = src;
    e->src_port = src_port;
    e->dst_node = dst;
    e->dst_port = dst_port;
    EdgeID id = edges.insert(std::move(e));
    edges.get(id)->id = id;
    existing_edges[key] = id;
    if (check_for_cycles && cycle_detected_()) {
        edges.remove(id);  // graph.rs:468 (the reference leaves the bookkeeping maps populated)
        return ERR_CYCLE_DETECTED;
    }
    needs_compile = true;
    return index_to_i64(id);
}

bool AudioGraph::disconnect(NodeID src, uint32_t src_port, NodeID dst, uint32_t dst_port) {
    // graph.rs:483-501
    auto key = std::make_tuple(index_to_i64(src), src_port, index_to_i64(dst), dst_port);
    auto it = existing_edges.find(key);
    if (it == existing_edges.end()) return false;
    EdgeID id = it->second;
    existing_edges.erase(it);
    disconnect_by_edge_id(id);
    return true;
}

bool AudioGraph::disconnect_by_edge_id(EdgeID id) {
    // graph.rs:507-524
    std::unique_ptr<Edge> e = edges.remove(id);
    if (!e) return false;
    existing_edges.erase(std::make_tuple(index_to_i64(e->src_node), e->src_port, index_to_i64(e->dst_node),
                                         e->dst_port));
    connected_input_ports.erase({index_to_i64(e->dst_node), e->dst_port});
    needs_compile = true;
    return true;
}

std::vector<EdgeID> AudioGraph::remove_edges_with_input_port(NodeID n, uint32_t port) {
    std::vector<EdgeID> rm;
    edges.for_each([&](Index i, Edge& e) {
        if (e.dst_node == n && e.dst_port == port) rm.push_back(i);
    });
    for (EdgeID i : rm) disconnect_by_edge_id(i);
    return rm;
}
std::vector<EdgeID> AudioGraph::remove_edges_with_output_port(NodeID n, uint32_t port) {
    std::vector<EdgeID> rm;
    edges.for_each([&](Index i, Edge& e) {
        if (e.src_node == n && e.src_port == port) rm.push_back(i);
    });
    for (EdgeID i : rm) disconnect_by_edge_id(i);
    return rm;
}

bool AudioGraph::cycle_detected_() { return cycle_detected(nodes, edges, graph_in_id, graph_out_id); }

// ================================================================= graph/processor.rs
int FirewheelProcessor::process_interleaved(const float* input, size_t input_len, float* output, size_t output_len,
                                            size_t num_in_channels, size_t num_out_channels, size_t frames,
                                            double stream_time_secs, uint32_t stream_status) {
    // processor.rs:61-165
    if (!schedule || frames == 0) {  // :86-89 (Q19)
        for (size_t i = 0; i < output_len; ++i) output[i] = 0.0f;
        return 0;
    }
    assert(input_len == frames * num_in_channels);    // :91
    assert(output_len == frames * num_out_channels);  // :92
    size_t frames_processed = 0;
    while (frames_processed < frames) {
        size_t block_frames = std::min(frames - frames_processed, max_block_frames);
        schedule->prepare_graph_inputs(
            block_frames, num_in_channels, [&](float* const* channels, size_t n) -> SilenceMask {
                return deinterleave(channels, n, block_frames, input + frames_processed * num_in_channels,
                                    block_frames * num_in_channels, num_in_channels, true);
            });
        process_block(block_frames, stream_time_secs, stream_status);  // Q18: same stream_time for sub-blocks
        schedule->read_graph_outputs(
            block_frames, num_out_channels, [&](const float* const* channels, size_t n, SilenceMask mask) {
                if (record_out_masks) record_out_masks->push_back(mask.bits | (n < 64 ? ~0ull << n : 0ull));  // channels past n are zero-filled
                float* dst = output + frames_processed * num_out_channels;
                size_t dst_len = block_frames * num_out_channels;
                if (n == 2 && num_out_channels == 2)
                    interleave_stereo(channels[0], channels[1], dst, dst_len, &mask);
                else
                    interleave(channels, n, block_frames, dst, dst_len, num_out_channels, &mask);
            });
        frames_processed += block_frames;
    }
    return 0;
}

void FirewheelProcessor::process_block(size_t block_frames, double stream_time_secs, uint32_t stream_status) {
    // processor.rs:208-248
    if (!schedule) return;
    schedule->process(block_frames, [&](NodeID id, SilenceMask in_mask, const float* const* inputs, size_t n_in,
                                        float* const* outputs, size_t n_out) -> SilenceMask {
        SilenceMask out_mask;  // NONE_SILENT (:233)
        ProcInfo info{in_mask, &out_mask, stream_time_secs, stream_status};
        nodes[id.slot]->process(block_frames, inputs, n_in, outputs, n_out, info);
        return out_mask;
    });
}

// ================================================================= graph/context.rs:93-137 + graph.rs:586-627
int Ctx::update() {
    if (!graph.needs_compile) return 0;
    std::unique_ptr<CompiledSchedule> sched;
    int rc = compile(graph.nodes, graph.edges, graph.graph_in_id, graph.graph_out_id, max_block_frames, sched);
    if (rc != 0) {
        last_error = rc == ERR_COMPILE_CYCLE ? "cycle detected" : "many-to-one";
        return rc;
    }
    std::vector<std::pair<NodeID, std::unique_ptr<AudioNodeProcessor>>> new_procs;
    for (NodeID id : graph.nodes_to_activate) {
        NodeEntry* ne = graph.nodes.get(id);
        if (!ne) continue;
        std::string err;
        auto p = ne->node->activate(sample_rate, max_block_frames, ne->num_inputs, ne->num_outputs, err);
        if (!p) {
            last_error = err;  // graph.rs:603-609: already-activated processors are handed back (dropped here)
            return ERR_COMPILE_NODE_ACTIVATION_FAILED;
        }
        new_procs.emplace_back(id, std::move(p));
    }
    // processor.rs:167-199 (poll_messages at the next block start)
    for (NodeID id : graph.nodes_to_remove_from_schedule) processor.nodes.erase(id.slot);
    for (auto& np : new_procs) processor.nodes[np.first.slot] = std::move(np.second);
    processor.schedule = std::move(sched);
    graph.needs_compile = false;
    graph.nodes_to_activate.clear();
    graph.nodes_to_remove_from_schedule.clear();
    return 0;
}

}  // namespace fwo

// ================================================================= flat C API (ctypes)
using namespace fwo;
extern "C" {

void* fwo_ctx_new(uint32_t sample_rate, uint32_t max_block_frames, uint32_t n_in, uint32_t n_out) {
    return new Ctx(sample_rate, max_block_frames, n_in, n_out);
}
void fwo_ctx_free(void* c) { delete (Ctx*)c; }
const char* fwo_last_error(void* c) { return ((Ctx*)c)->last_error.c_str(); }
int64_t fwo_graph_in_node(void* c) { return index_to_i64(((Ctx*)c)->graph.graph_in_id); }
int64_t fwo_graph_out_node(void* c) { return index_to_i64(((Ctx*)c)->graph.graph_out_id); }

int64_t fwo_add_node(void* c, int kind, uint32_t n_in, uint32_t n_out, const float* params, int n_params) {
    Ctx* cx = (Ctx*)c;
    auto node = make_node(kind, params, n_params);
    if (kind == KIND_FIR || kind == KIND_RESAMPLER) {
        int id = n_params > 0 ? (int)params[0] : -1;
        if (id < 0 || id >= (int)cx->samples.size()) return -20;
        node->ir = cx->samples[id];
    }
    return index_to_i64(cx->graph.add_node(n_in, n_out, std::move(node)));
}
int fwo_custom_node_set_process(void* c, int64_t node, AudioNode::CustomFn fn, void* user) {
    Ctx* cx = (Ctx*)c;
    auto* ne = cx->graph.nodes.get(index_from_i64(node));
    if (!ne || !ne->node || ne->node->kind != KIND_CUSTOM) return -20;
    ne->node->custom_fn = fn;
    ne->node->custom_user = user;
    return 0;
}
int fwo_remove_node(void* c, int64_t node) { return ((Ctx*)c)->graph.remove_node(index_from_i64(node)); }
int64_t fwo_connect(void* c, int64_t src, uint32_t sp, int64_t dst, uint32_t dp, int check) {
    return ((Ctx*)c)->graph.connect(index_from_i64(src), sp, index_from_i64(dst), dp, check != 0);
}
int fwo_disconnect(void* c, int64_t src, uint32_t sp, int64_t dst, uint32_t dp) {
    return ((Ctx*)c)->graph.disconnect(index_from_i64(src), sp, index_from_i64(dst), dp) ? 1 : 0;
}
int fwo_disconnect_edge(void* c, int64_t e) {
    return ((Ctx*)c)->graph.disconnect_by_edge_id(index_from_i64(e)) ? 1 : 0;
}
int fwo_cycle_detected(void* c) { return ((Ctx*)c)->graph.cycle_detected_() ? 1 : 0; }
int fwo_update(void* c) { return ((Ctx*)c)->update(); }

int fwo_sched_len(void* c) {
    Ctx* cx = (Ctx*)c;
    return cx->processor.schedule ? (int)cx->processor.schedule->schedule.size() : -1;
}
int fwo_sched_num_buffers(void* c) { return (int)((Ctx*)c)->processor.schedule->num_buffers; }
int64_t fwo_sched_node(void* c, int i) { return index_to_i64(((Ctx*)c)->processor.schedule->schedule[i].id); }
int fwo_sched_in(void* c, int i, int* buf, int* clear, int cap) {
    auto& sn = ((Ctx*)c)->processor.schedule->schedule[i];
    int n = (int)sn.input_buffers.size();
    for (int k = 0; k < n && k < cap; ++k) {
        buf[k] = (int)sn.input_buffers[k].buffer_index;
        clear[k] = sn.input_buffers[k].should_clear ? 1 : 0;
    }
    return n;
}
int fwo_sched_out(void* c, int i, int* buf, int cap) {
    auto& sn = ((Ctx*)c)->processor.schedule->schedule[i];
    int n = (int)sn.output_buffers.size();
    for (int k = 0; k < n && k < cap; ++k) buf[k] = (int)sn.output_buffers[k].buffer_index;
    return n;
}

// samples: interleaved formats take frames*channels items; planar formats take channels planes of `frames`
// items concatenated ([ch][frame]).
int fwo_sample_new(void* c, int format, uint32_t channels, uint64_t frames, const void* data) {
    Ctx* cx = (Ctx*)c;
    auto s = std::make_shared<SampleResource>();
    s->format = format;
    s->channels = channels;
    s->frames = frames;
    size_t n = (size_t)(frames * channels);
    switch (format) {
        case FMT_INTERLEAVED_I16:
        case FMT_PLANAR_I16:
            s->i16.assign((const int16_t*)data, (const int16_t*)data + n);
            break;
        case FMT_INTERLEAVED_U16:
        case FMT_PLANAR_U16:
            s->u16.assign((const uint16_t*)data, (const uint16_t*)data + n);
            break;
        default:
            s->f32.assign((const float*)data, (const float*)data + n);
            break;
    }
    cx->samples.push_back(s);
    return (int)cx->samples.size() - 1;
}

// runtime params.  VOLUME/SAMPLER: 0 = percent_volume.  BEEP: 0 = enabled.  PAN: 0 = pan.
int fwo_set_param(void* c, int64_t node, int param, float value) {
    Ctx* cx = (Ctx*)c;
    NodeEntry* ne = cx->graph.nodes.get(index_from_i64(node));
    if (!ne) return -1;
    AudioNode* n = ne->node.get();
    switch (n->kind) {
        case KIND_VOLUME:
        case KIND_SAMPLER:
            if (param != 0) return -2;
            *n->raw_gain = percent_volume_to_raw_gain(value);  // volume.rs:28-34, sampler.rs:171-177
            return 0;
        case KIND_BEEP_TEST:
            if (param != 0) return -2;
            *n->enabled = value != 0.0f;
            return 0;
        case KIND_STEREO_PAN: {
            if (param != 0) return -2;
            fwo::pan_to_gains(value, n->aux0.get(), n->aux1.get());
            return 0;
        }
        case KIND_STEREO_WIDTH:
            if (param != 0) return -2;
            *n->aux0 = fmaxf(value, 0.0f);
            return 0;
        case KIND_BIQUAD:
            if (param != 1 && param != 2) return -2;
            n->spec_params[param] = value;
            fwo::biquad_coefs((int)n->spec_params[0], n->spec_params[1], n->spec_params[2], n->act_sample_rate,
                              n->coefs->data());
            return 0;
        case KIND_DELAY:
            if (param == 1) {
                *n->raw_gain = fminf(fmaxf(value, 0.0f), 0.999f);
                return 0;
            }
            if (param != 2) return -2;
            *n->aux0 = fminf(fmaxf(value, 0.0f), 1.0f);
            *n->aux1 = 1.0f - *n->aux0;
            return 0;
        case KIND_RESAMPLER:  // 1 = ratio, 3 = playing, 4 = seek to source frame
            if (param == 1) (*n->ctl)[0] = (double)fwo::resampler_step(value);
            else if (param == 3) (*n->ctl)[1] = value != 0.0f ? 1.0 : 0.0;
            else if (param == 4) {
                (*n->ctl)[3] = 1.0;
                (*n->ctl)[4] = (double)(uint64_t)fmaxf(value, 0.0f);
            } else return -2;
            return 0;
        case KIND_SPATIAL: {  // 0/1/2 = x/y/z of the source relative to the listener
            if (param < 0 || param > 2) return -2;
            n->spec_params[param] = value;
            int dl, dr;
            fwo::spatial_params(n->spec_params[0], n->spec_params[1], n->spec_params[2], n->act_sample_rate, n->aux0.get(),
                                n->aux1.get(), &dl, &dr);
            (*n->ctl)[0] = dl;
            (*n->ctl)[1] = dr;
            return 0;
        }
        default:
            return -2;
    }
}

static int push_msg(void* c, int64_t node, SamplerMsg m) {
    Ctx* cx = (Ctx*)c;
    NodeEntry* ne = cx->graph.nodes.get(index_from_i64(node));
    if (!ne || ne->node->kind != KIND_SAMPLER) return -1;
    if (ne->node->to_processor->size() >= 128) return -3;  // rtrb capacity (sampler.rs:14)
    ne->node->to_processor->push_back(m);
    return 0;
}
int fwo_sampler_set_sample(void* c, int64_t node, int sample, int stop_playback) {
    Ctx* cx = (Ctx*)c;
    if (sample < 0 || sample >= (int)cx->samples.size()) return -1;
    SamplerMsg m;
    m.type = SamplerMsg::SetSample;
    m.sample = cx->samples[sample];
    m.stop_playback = stop_playback != 0;
    return push_msg(c, node, m);
}
int fwo_sampler_play(void* c, int64_t node) {
    SamplerMsg m;
    m.type = SamplerMsg::Play;
    return push_msg(c, node, m);
}
int fwo_sampler_pause(void* c, int64_t node) {
    SamplerMsg m;
    m.type = SamplerMsg::Pause;
    return push_msg(c, node, m);
}
int fwo_sampler_stop(void* c, int64_t node) {
    SamplerMsg m;
    m.type = SamplerMsg::Stop;
    return push_msg(c, node, m);
}
int fwo_sampler_set_playhead_secs(void* c, int64_t node, double secs) {
    SamplerMsg m;
    m.type = SamplerMsg::SetPlayheadSecs;
    m.playhead_secs = secs;
    return push_msg(c, node, m);
}
int fwo_sampler_set_loop_range(void* c, int64_t node, int mode, double start, double end) {
    SamplerMsg m;
    m.type = SamplerMsg::SetLoopRange;
    m.loop_mode = mode;
    m.loop_start = start;
    m.loop_end = end;
    return push_msg(c, node, m);
}

int fwo_process_interleaved(void* c, const float* in, float* out, uint32_t n_in_ch, uint32_t n_out_ch,
                            uint64_t frames, double t, uint32_t status) {
    Ctx* cx = (Ctx*)c;
    return cx->processor.process_interleaved(in, (size_t)frames * n_in_ch, out, (size_t)frames * n_out_ch, n_in_ch,
                                             n_out_ch, (size_t)frames, t, status);
}

// ... also reporting read_graph_outputs' silence mask of every block (bit c = output channel c silent; test hook)
int fwo_process_interleaved_masks(void* c, const float* in, float* out, uint32_t n_in_ch, uint32_t n_out_ch, uint64_t frames,
                                  double t, uint32_t status, uint64_t* masks, uint32_t cap) {
    Ctx* cx = (Ctx*)c;
    std::vector<uint64_t> rec;
    cx->processor.record_out_masks = &rec;
    int rc = cx->processor.process_interleaved(in, (size_t)frames * n_in_ch, out, (size_t)frames * n_out_ch, n_in_ch, n_out_ch,
                                               (size_t)frames, t, status);
    cx->processor.record_out_masks = nullptr;
    for (size_t i = 0; i < rec.size() && i < cap; ++i) masks[i] = rec[i];
    return rc ? rc : (int)rec.size();
}

// cpu_baseline "all cores" (the generous baseline): n independent engines — the voices of one graph split over
// them — each driven by its own std::thread for `secs` seconds, calls of `frames_per_call` frames back to back into a scratch
// buffer, no mix-bus exchange charged.  blocks_done[i] = process calls engine i completed.  Returns the wall seconds.
double fwo_process_parallel(void* const* ctxs, int n, uint32_t n_out_ch, uint64_t frames_per_call, double secs, uint64_t* calls_done) {
    std::atomic<int> ready{0};
    std::atomic<bool> go{false}, stop{false};
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i) {
        calls_done[i] = 0;
        th.emplace_back([&, i] {
            Ctx* cx = (Ctx*)ctxs[i];
            std::vector<float> out((size_t)frames_per_call * n_out_ch);
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            while (!stop.load(std::memory_order_relaxed)) {
                cx->processor.process_interleaved(nullptr, 0, out.data(), out.size(), 0, n_out_ch, (size_t)frames_per_call, 0.0, 0);
                calls_done[i]++;
            }
        });
    }
    while (ready.load() < n) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    std::this_thread::sleep_for(std::chrono::duration<double>(secs));
    stop.store(true);
    for (auto& t : th) t.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// B1-level: call one activated node's process() directly with caller buffers (processor.rs:243).
int fwo_node_process(void* c, int64_t node, uint64_t frames, const float* const* inputs, uint32_t n_in,
                     float* const* outputs, uint32_t n_out, uint64_t in_mask, uint64_t* out_mask, double t,
                     uint32_t status) {
    Ctx* cx = (Ctx*)c;
    Index id = index_from_i64(node);
    auto it = cx->processor.nodes.find(id.slot);
    if (it == cx->processor.nodes.end()) return -1;
    SilenceMask im, om;
    im.bits = in_mask;
    om.bits = *out_mask;
    ProcInfo info{im, &om, t, status};
    it->second->process((size_t)frames, inputs, n_in, outputs, n_out, info);
    *out_mask = om.bits;
    return 0;
}

// ---- primitive-level entry points (Appendix A unit tests)
void* fwo_smoother_new(float val, uint32_t sr, uint32_t mbf) { return new ParamSmoother(val, sr, mbf); }
void fwo_smoother_free(void* s) { delete (ParamSmoother*)s; }
void fwo_smoother_set(void* s, float v) { ((ParamSmoother*)s)->set(v); }
void fwo_smoother_reset(void* s, float v) { ((ParamSmoother*)s)->reset(v); }
// returns the length of the returned slice; copies min(len, cap) values into out; *status gets the status
int fwo_smoother_process(void* s, uint32_t frames, float* out, uint32_t cap, int* status) {
    SmoothedOutput o = ((ParamSmoother*)s)->process(frames);
    for (size_t i = 0; i < o.len && i < cap; ++i) out[i] = o.values[i];
    *status = (int)o.status;
    return (int)o.len;
}
void fwo_smoother_state(void* s, float* input, float* last_output, float* a, float* b, int* status) {
    ParamSmoother* p = (ParamSmoother*)s;
    *input = p->input;
    *last_output = p->last_output;
    *a = p->a;
    *b = p->b;
    *status = (int)p->status;
}
uint64_t fwo_mask_new_all_silent(uint64_t n) { return SilenceMask::new_all_silent((size_t)n).bits; }
int fwo_mask_any_silent(uint64_t m, uint64_t n) {
    SilenceMask s;
    s.bits = m;
    return s.any_channel_silent((size_t)n);
}
int fwo_mask_all_silent(uint64_t m, uint64_t n) {
    SilenceMask s;
    s.bits = m;
    return s.all_channels_silent((size_t)n);
}
float fwo_db_to_gain(float db) { return db_to_gain(db); }
float fwo_db_to_gain_clamped(float db) { return db_to_gain_clamped_neg_100_db(db); }
float fwo_gain_to_db_clamped(float a) { return gain_to_db_clamped_neg_100_db(a); }
float fwo_percent_volume_to_raw_gain(float p) { return percent_volume_to_raw_gain(p); }
float fwo_pcm_i16_to_f32(int16_t s) { return pcm_i16_to_f32(s); }
float fwo_pcm_u16_to_f32(uint16_t s) { return pcm_u16_to_f32(s); }
void fwo_pan_to_gains(float pan, float* gl, float* gr) { fwo::pan_to_gains(pan, gl, gr); }

uint64_t fwo_deinterleave(float* const* channels, uint32_t n_channels, uint32_t ch_len, const float* interleaved,
                          uint32_t interleaved_len, uint32_t nic, int calc) {
    return deinterleave(channels, n_channels, ch_len, interleaved, interleaved_len, nic, calc != 0).bits;
}
void fwo_interleave(const float* const* channels, uint32_t n_channels, uint32_t ch_len, float* interleaved,
                    uint32_t interleaved_len, uint32_t nic, int has_mask, uint64_t mask) {
    SilenceMask m;
    m.bits = mask;
    interleave(channels, n_channels, ch_len, interleaved, interleaved_len, nic, has_mask ? &m : nullptr);
}
void fwo_interleave_stereo(const float* l, const float* r, float* interleaved, uint32_t interleaved_len,
                           int has_mask, uint64_t mask) {
    SilenceMask m;
    m.bits = mask;
    interleave_stereo(l, r, interleaved, interleaved_len, has_mask ? &m : nullptr);
}


// ---- firewheel-cpal/src/lib.rs:362-449 DataCallback: the backend callback's stream-time / underflow bookkeeping,
// with cpal's `info.timestamp().callback` instant replaced by a caller-supplied clock value in seconds (the only thing
// the code does with the instant is subtract the first one from it, :402-407).
struct FwoStream {
    void* ctx;
    uint32_t num_in_channels, num_out_channels;
    double sample_rate_recip;      // :371
    bool has_first_instant;        // first_stream_instant: Option<StreamInstant> :372
    double first_stream_instant;
    double predicted_stream_secs;  // :373 (1.0)
    bool is_first_callback;        // :374
};
void* fwo_stream_new(void* c, uint32_t sample_rate, uint32_t n_in_ch, uint32_t n_out_ch) {
    FwoStream* s = new FwoStream;
    s->ctx = c;
    s->num_in_channels = n_in_ch;
    s->num_out_channels = n_out_ch;
    s->sample_rate_recip = 1.0 / (double)sample_rate;  // f64::from(sample_rate).recip()
    s->has_first_instant = false;
    s->first_stream_instant = 0.0;
    s->predicted_stream_secs = 1.0;
    s->is_first_callback = true;
    return s;
}
void fwo_stream_free(void* s) { delete (FwoStream*)s; }
// returns the StreamStatus bits passed to process_interleaved; *stream_time_out = the stream_time_secs passed
int fwo_stream_callback(void* sp, float* output, uint64_t frames, double callback_instant_secs, double* stream_time_out) {
    FwoStream* s = (FwoStream*)sp;
    double stream_time_secs;
    bool underflow;
    if (s->is_first_callback) {  // :385-393
        s->is_first_callback = false;
        s->predicted_stream_secs = (double)frames * s->sample_rate_recip;
        stream_time_secs = 0.0;
        underflow = false;
    } else if (s->has_first_instant) {  // :394-413
        stream_time_secs = callback_instant_secs - s->first_stream_instant;
        const bool underrun = stream_time_secs > s->predicted_stream_secs;  // :405
        s->predicted_stream_secs = stream_time_secs + ((double)frames * s->sample_rate_recip * 1.2);  // :411-412
        underflow = underrun;
    } else {  // :414-419
        s->has_first_instant = true;
        s->first_stream_instant = callback_instant_secs;
        stream_time_secs = s->predicted_stream_secs;
        s->predicted_stream_secs += (double)frames * s->sample_rate_recip * 1.2;
        underflow = false;
    }
    uint32_t stream_status = 0;           // StreamStatus::empty() :423
    if (underflow) stream_status |= 0b10;  // OUTPUT_UNDERFLOW :425-427, core/node.rs:130
    if (stream_time_out) *stream_time_out = stream_time_secs;
    // :429-437 (no processor yet -> output.fill(0.0) :442-445 is what process_interleaved does without a schedule, Q19)
    fwo_process_interleaved(s->ctx, nullptr, output, s->num_in_channels, s->num_out_channels, frames, stream_time_secs,
                            stream_status);
    return (int)stream_status;
}

}  // extern "C"
