// fwgpu_control_math.cpp — control-half scalar math, done on the host exactly as the reference's control thread does it,
// and the initial audio-half state of every node kind (the AudioNode constructors + activate()).
#include "fwgpu_ctx.h"

namespace fwgpu {

// ---- control-half scalar math, done on the host exactly as the reference's control thread does it
float percent_volume_to_raw_gain(float p) {  // core/param/range.rs:32-35
    float n = fmaxf(p, 0.0f) * (1.0f / 100.0f);
    return n * n;
}
float db_to_gain_clamped_neg_100_db(float db) {  // core/util.rs:7-9,21-27
    if (db <= -100.0f) return 0.0f;
    return powf(10.0f, 0.05f * db);
}
void pan_to_gains(float pan, float* gl, float* gr) {  // SPEC: DESIGN.md "spec nodes / pan"
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
Smoother make_smoother(float val, uint32_t sample_rate) {  // core/param/smoother.rs:93-112, defaults :18-25
    Smoother s;
    const float smooth_secs = 10.0f / 1000.0f;
    s.b = expf(-1.0f / (smooth_secs * (float)sample_rate));
    s.a = 1.0f - s.b;
    s.status = SM_INACTIVE;
    s.input = val;
    s.last = val;
    s.eps = 0.00001f;
    return s;
}

// SPEC biquad: RBJ cookbook, computed in f64 on the control side, normalised by a0, rounded to f32.
void biquad_coefs(int type, float cutoff_hz, float q, uint32_t sample_rate, float co[5]) {
    double fs = (double)sample_rate;
    double f0 = fmin(fmax((double)cutoff_hz, 1.0), 0.49 * fs);
    double Q = fmax((double)q, 1e-3);
    double w0 = 2.0 * 3.14159265358979323846 * f0 / fs;
    double cw = cos(w0), alpha = sin(w0) / (2.0 * Q);
    double b0, b1, b2, a0 = 1.0 + alpha, a1 = -2.0 * cw, a2 = 1.0 - alpha;
    if (type == 1) {  // high-pass
        b0 = (1.0 + cw) * 0.5;
        b1 = -(1.0 + cw);
        b2 = (1.0 + cw) * 0.5;
    } else if (type == 2) {  // band-pass, constant 0 dB peak gain
        b0 = alpha;
        b1 = 0.0;
        b2 = -alpha;
    } else {  // low-pass
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
// SPEC resampler: Kaiser-windowed sinc (beta 8, cutoff 0.9 x Nyquist), RS_PHASES x RS_TAPS, every
// phase normalised to unity DC gain in f64 and rounded to f32 — the control side builds the table once per ctx.
double bessel_i0(double x) {
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
            double t = (double)(k - (RS_TAPS / 2 - 1)) - (double)ph / RS_PHASES;
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
uint64_t resampler_step(float ratio) {  // source frames per output frame as 32.32 fixed point
    double r = (double)ratio;
    if (!(r >= 1.0 / 256.0)) r = 1.0 / 256.0;
    if (r > 256.0) r = 256.0;
    return (uint64_t)llround(r * 4294967296.0);
}
// SPEC spatialiser: listener at the origin (+x right, +y up, -z forward): inverse-distance gain (reference distance
// 1, rolloff 1), equal-power pan from the direction cosine to the right, per-ear delay up to 0.66 ms.
void spatial_params(float x, float y, float z, uint32_t sample_rate, float* gl, float* gr, int* dl, int* dr) {
    const double pi = 3.14159265358979323846;
    double d = sqrt((double)x * x + (double)y * y + (double)z * z);
    double att = 1.0 / fmax(d, 1.0);
    double s = d < 1e-9 ? 0.0 : (double)x / d;
    double theta = (s + 1.0) * (pi / 4.0);
    *gl = (float)(cos(theta) * att);
    *gr = (float)(sin(theta) * att);
    double itd_max = round(0.00066 * (double)sample_rate);
    if (itd_max > SP_HIST - 1) itd_max = SP_HIST - 1;
    *dl = (int)round(fmax(0.0, s) * itd_max);
    *dr = (int)round(fmax(0.0, -s) * itd_max);
}
uint32_t delay_frames(float secs, uint32_t sample_rate) {
    double d = round((double)secs * (double)sample_rate);
    if (!(d >= 1.0)) d = 1.0;
    if (d > 16777216.0) d = 16777216.0;
    return (uint32_t)d;
}

// AudioNode constructors + activate(): the initial audio-half state of each node kind.
NodeState make_state(int kind, const float* params, int n_params, uint32_t sample_rate) {
    auto p = [&](int i, float d) { return i < n_params ? params[i] : d; };
    NodeState s;
    memset(&s, 0, sizeof(s));
    s.sample = -1;
    s.sample_rate = sample_rate;
    s.s0 = make_smoother(0.f, sample_rate);
    s.s1 = make_smoother(0.f, sample_rate);
    switch (kind) {
        case K_VOLUME:   // volume.rs:15-22, :67-75
        case K_SAMPLER:  // sampler.rs:55-64, :302-319
            s.p0 = percent_volume_to_raw_gain(fmaxf(p(0, 100.0f), 0.0f));
            s.s0 = make_smoother(s.p0, sample_rate);
            break;
        case K_BEEP: {  // beep_test.rs:15-24, :55-60
            float f = p(0, 440.0f);
            if (f < 20.0f) f = 20.0f;
            if (f > 20000.0f) f = 20000.0f;
            float g = db_to_gain_clamped_neg_100_db(p(1, -12.0f));
            if (g < 0.0f) g = 0.0f;
            if (g > 1.0f) g = 1.0f;
            s.gain = g;
            s.enabled = p(2, 1.0f) != 0.0f ? 1 : 0;
            s.phasor = 0.0f;
            s.phasor_inc = f / (float)sample_rate;
            break;
        }
        case K_HARD_CLIP:  // hard_clip.rs:8-12
            s.p0 = db_to_gain_clamped_neg_100_db(p(0, 0.0f));
            break;
        case K_PAN:
            pan_to_gains(p(0, 0.0f), &s.p0, &s.p1);
            s.s0 = make_smoother(s.p0, sample_rate);
            s.s1 = make_smoother(s.p1, sample_rate);
            break;
        case K_WIDTH:
            s.p0 = fmaxf(p(0, 1.0f), 0.0f);
            s.s0 = make_smoother(s.p0, sample_rate);
            break;
        case K_BIQUAD:  // coefficients go to the ext pool at activation; keep the ctor args for that
            s.p0 = p(1, 1000.0f);  // cutoff
            s.p1 = p(2, 0.70710678f);  // Q
            s.enabled = (int)p(0, 0.0f);  // type
            break;
        case K_FIR:
            s.sample = (int)p(0, -1.0f);  // impulse-response sample id; T and the ring are set at activation
            break;
        case K_RESAMPLER:  // params: sample id, ratio, loop, playing
            s.sample = (int)p(0, -1.0f);
            s.loop_start = resampler_step(p(1, 1.0f));
            s.has_loop = p(2, 0.0f) != 0.0f ? 1 : 0;
            s.playing = p(3, 1.0f) != 0.0f ? 1 : 0;
            s.playhead = 0;
            break;
        case K_SPATIAL: {  // params: x, y, z of the source; the ctor args stay in phasor / phasor_inc / gain
            s.phasor = p(0, 0.0f);
            s.phasor_inc = p(1, 0.0f);
            s.gain = p(2, -1.0f);
            int dl, dr;
            spatial_params(s.phasor, s.phasor_inc, s.gain, sample_rate, &s.p0, &s.p1, &dl, &dr);
            s.s0 = make_smoother(s.p0, sample_rate);
            s.s1 = make_smoother(s.p1, sample_rate);
            s.playing = dl;
            s.has_loop = dr;
            break;
        }
        case K_DELAY: {
            float mix = fminf(fmaxf(p(2, 0.5f), 0.0f), 1.0f);
            s.p0 = fminf(fmaxf(p(1, 0.0f), 0.0f), 0.999f);  // feedback
            s.p1 = mix;
            s.gain = 1.0f - mix;  // dry
            s.loop_end = delay_frames(p(0, 0.1f), sample_rate);
            s.playhead = 0;
            break;
        }
        default:
            break;
    }
    return s;
}

}  // namespace fwgpu
