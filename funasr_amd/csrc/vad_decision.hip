// Host-side decision logic of FSMN-VAD in native code (no device work in this file): frame classification from the
// silence posterior and the frame energy, the sliding-window sil<->speech detector, the start / end point state machine
// with look-back / look-ahead / maximum segment length / end-silence timeout, history dropping and the two reporting
// conventions of funasr/models/fsmn_vad_streaming/model.py (GetFrameState :761-823, WindowDetector :218-320,
// DetectOneFrame :1158-1302, On*/Pop* :552-736, DropCachedFrames :406-433, forward :861-905).
// The readable restatement is funasr_amd/vad_decision.py (same structure, same names); both are pinned to the reference's
// own class by tests/golden/vad_decision.npz. This version exists because the per-frame loop in Python handles ~4 000 x
// real time per host core, five times less than one GPU decodes.
#include "common.h"

#include <cmath>
#include <deque>
#include <memory>
#include <vector>

#include "../../include/paraformer_hip.h"

namespace pf {
namespace {

enum { NOT_STARTED = 1, IN_SPEECH = 2, ENDED = 3 };
enum Change { SIL_TO_SPEECH, SPEECH_TO_SIL, STAY_SPEECH, STAY_SIL };

struct Segment { int start_ms, end_ms; bool has_start, has_end; };

struct Decision {
    pf_vad_options o;
    int shift, flen;
    // window detector
    std::vector<int> ring; int pos = 0, total = 0; bool speaking = false; int win_size, to_speech, to_sil;
    double max_end_sil_ms, speech_noise_thres;
    int state = NOT_STARTED;
    long n_frames = 0, dropped = 0, buf_start = 0;
    std::deque<double> sil_score, decibel;
    double noise_db = -100.0;
    long kept_samples = 0, buf_off = 0;
    bool first_block = true;
    long last_speech = 0, last_silence = -1, sil_run = 0, start_frame = -1, end_frame = -1, n_ends = 0;
    std::vector<Segment> segments;
    size_t reported = 0;
    bool next_is_new = true;
    std::string error;

    explicit Decision(const pf_vad_options& opt) : o(opt) {
        shift = (int)(o.frame_in_ms * o.sample_rate / 1000);
        flen = (int)(o.frame_length_ms * o.sample_rate / 1000);
        win_size = (int)(o.window_size_ms / o.frame_in_ms);
        to_speech = (int)(o.sil_to_speech_time_thres / o.frame_in_ms);
        to_sil = (int)(o.speech_to_sil_time_thres / o.frame_in_ms);
        ring.assign(win_size > 0 ? win_size : 1, 0);
        max_end_sil_ms = o.max_end_silence_time - o.speech_to_sil_time_thres;
        speech_noise_thres = o.speech_noise_thres;
    }
    void win_reset() { std::fill(ring.begin(), ring.end(), 0); pos = 0; total = 0; speaking = false; }
    Change win_push(int label) {
        total += label - ring[pos];
        ring[pos] = label;
        pos = (pos + 1) % win_size;
        if (!speaking && total >= to_speech) { speaking = true; return SIL_TO_SPEECH; }
        if (speaking && total <= to_sil) { speaking = false; return SPEECH_TO_SIL; }
        return speaking ? STAY_SPEECH : STAY_SIL;
    }
    long buf_len() const { return kept_samples - buf_off > 0 ? kept_samples - buf_off : 0; }
    void reslice() { buf_off = (buf_start - dropped > 0 ? buf_start - dropped : 0) * shift; }
    void take_samples(long n) {
        if (first_block) { kept_samples = (n - 1) * shift + flen; first_block = false; }
        else kept_samples += n * shift;
        reslice();
    }
    bool drop_before(long frame) {
        if (frame > n_frames) { error = "cannot drop history beyond the frames seen"; return false; }
        const long k = frame - dropped;
        if (k <= 0) return true;
        kept_samples = kept_samples - k * shift > 0 ? kept_samples - k * shift : 0;
        const long kk = k < (long)decibel.size() ? k : (long)decibel.size();
        decibel.erase(decibel.begin(), decibel.begin() + kk);
        sil_score.erase(sil_score.begin(), sil_score.begin() + kk);
        dropped = frame;
        reslice();
        return true;
    }
    bool skip_to(long frame) {
        while (buf_start < frame) {
            if (buf_len() < shift) { error = "waveform history exhausted while advancing the segment buffer"; return false; }
            ++buf_start;
            buf_off = (buf_start - dropped) * shift;
        }
        return true;
    }
    bool emit(long frame, long count, bool opens, bool closes) {
        if (!skip_to(frame)) return false;
        if (segments.empty() || opens) segments.push_back(Segment{(int)(frame * o.frame_in_ms), (int)(frame * o.frame_in_ms), false, false});
        Segment& s = segments.back();
        buf_start += count;
        s.end_ms = (int)((frame + count) * o.frame_in_ms);
        s.has_start = s.has_start || opens;
        s.has_end = s.has_end || closes;
        return true;
    }
    bool speech(long frame) { last_speech = frame; return emit(frame, 1, false, false); }
    bool silence(long frame) { last_silence = frame; return state == NOT_STARTED ? skip_to(frame) : true; }
    bool begin(long frame, bool fake) {
        if (start_frame == -1) start_frame = frame;
        if (!fake && state == NOT_STARTED) return emit(start_frame, 1, true, false);
        return true;
    }
    bool finish(long frame, bool fake) {
        for (long t = last_speech + 1; t < frame; ++t) if (!speech(t)) return false;
        if (end_frame == -1) end_frame = frame;
        if (!fake && !emit(end_frame, 1, false, true)) return false;
        ++n_ends;
        return true;
    }
    bool restart() {
        sil_run = 0; last_speech = 0; last_silence = -1; start_frame = end_frame = -1;
        state = NOT_STARTED;
        win_reset();
        if (!segments.empty()) {
            if (!segments.back().has_end) { error = "detection restarted on an open segment"; return false; }
            return drop_before((long)(segments.back().end_ms / o.frame_in_ms));
        }
        return true;
    }
    int label(long t) {
        if (t >= (long)decibel.size()) return 0;
        const double db = decibel[t], snr = db - noise_db;
        if (db < o.decibel_thres) return 0;
        const double p = sil_score[t];
        const double noise_prob = std::log(p) * o.speech_2_noise_ratio, speech_prob = std::log(1.0 - p);
        if (std::exp(speech_prob) >= std::exp(noise_prob) + speech_noise_thres)
            return (snr >= o.snr_thres && db >= o.decibel_thres) ? 1 : 0;
        if (noise_db < -99.9) noise_db = db;
        else noise_db = (db + noise_db * (o.noise_frame_num_used_for_snr - 1)) / o.noise_frame_num_used_for_snr;
        return 0;
    }
    long latency() const { return win_size + (o.do_extend ? (long)(o.lookback_time_start_point / o.frame_in_ms) : 0); }
    bool continue_or_close(long frame, bool final) {
        const bool too_long = (double)(frame - start_frame + 1) > (double)o.max_single_segment_time / o.frame_in_ms;
        if (too_long || final) { if (!finish(frame, false)) return false; state = ENDED; return true; }
        return speech(frame);
    }
    bool step(int lab, long frame, bool final) {
        const double ms = o.frame_in_ms;
        if (lab == 1 && !(std::fabs(1.0) > o.fe_prior_thres)) lab = 0;
        const Change ch = win_push(lab);
        if (ch == SIL_TO_SPEECH) {
            sil_run = 0;
            if (state == NOT_STARTED) {
                const long first = buf_start > frame - latency() ? buf_start : frame - latency();
                if (!begin(first, false)) return false;
                state = IN_SPEECH;
                for (long t = first + 1; t <= frame; ++t) if (!speech(t)) return false;
            } else if (state == IN_SPEECH) {
                for (long t = last_speech + 1; t < frame; ++t) if (!speech(t)) return false;
                if (!continue_or_close(frame, final)) return false;
            }
        } else if (ch == SPEECH_TO_SIL || ch == STAY_SPEECH) {
            sil_run = 0;
            if (state == IN_SPEECH && !continue_or_close(frame, final)) return false;
        } else {
            ++sil_run;
            if (state == NOT_STARTED) {
                const bool timed_out = o.detect_mode == 0 && sil_run * ms > o.max_start_silence_time;
                if (timed_out || (final && n_ends == 0)) {
                    for (long t = last_silence + 1; t < frame; ++t) if (!silence(t)) return false;
                    if (!begin(0, true) || !finish(0, true)) return false;
                    state = ENDED;
                } else if (frame >= latency()) {
                    if (!silence(frame - latency())) return false;
                }
            } else if (state == IN_SPEECH) {
                const bool too_long = (double)(frame - start_frame + 1) > (double)o.max_single_segment_time / ms;
                if (sil_run * ms >= max_end_sil_ms) {
                    long back = (long)(max_end_sil_ms / ms);
                    if (o.do_extend) { back = back - (long)(o.lookahead_time_end_point / ms) - 1; if (back < 0) back = 0; }
                    if (!finish(frame - back, false)) return false;
                    state = ENDED;
                } else if (too_long) {
                    if (!finish(frame, false)) return false;
                    state = ENDED;
                } else if (o.do_extend && !final) {
                    if (sil_run <= (long)(o.lookahead_time_end_point / ms) && !speech(frame)) return false;
                } else if (final) {
                    if (!finish(frame, false)) return false;
                    state = ENDED;
                }
            }
        }
        if (state == ENDED && o.detect_mode == 1) return restart();
        return true;
    }
    int push(const float* p_sil, const float* db, int n, bool is_final, bool events, int32_t* out, int cap) {
        if (n <= 0) return 0;
        take_samples(n);
        for (int i = 0; i < n; ++i) { decibel.push_back((double)db[i]); sil_score.push_back((double)p_sil[i]); }
        n_frames += n;
        if (state != ENDED)
            for (int back = n - 1; back >= 0; --back) {
                const long frame = n_frames - 1 - back;
                if (!step(label(frame - dropped), frame, is_final && back == 0)) return -1;
            }
        if (!drop_before(buf_start)) return -1;
        int m = 0;
        const size_t n_seg = segments.size();
        for (size_t i = reported; i < n_seg; ++i) {
            const Segment& s = segments[i];
            int beg, end;
            if (events) {
                if (!s.has_start || (!next_is_new && !s.has_end)) continue;
                beg = next_is_new ? s.start_ms : -1;
                if (s.has_end) { end = s.end_ms; next_is_new = true; ++reported; }
                else { end = -1; next_is_new = false; }
            } else {
                if (!is_final && !(s.has_start && s.has_end)) continue;
                beg = s.start_ms; end = s.end_ms;
                ++reported;
            }
            if (m < cap) { out[2 * m] = beg; out[2 * m + 1] = end; }
            ++m;
        }
        return m;
    }
};

}  // namespace
}  // namespace pf

extern "C" {

pf_vad_decision* pf_vad_decision_create(const pf_vad_options* opts) {
    if (!opts) { pf::set_error("vad_decision: null options"); return nullptr; }
    if (opts->frame_in_ms <= 0 || opts->window_size_ms < opts->frame_in_ms || opts->sample_rate <= 0) {
        pf::set_error("vad_decision: frame_in_ms > 0, window_size_ms >= frame_in_ms, sample_rate > 0");
        return nullptr;
    }
    return reinterpret_cast<pf_vad_decision*>(new pf::Decision(*opts));
}
void pf_vad_decision_destroy(pf_vad_decision* d) { delete reinterpret_cast<pf::Decision*>(d); }
void pf_vad_decision_set_thresholds(pf_vad_decision* dh, double max_end_sil_ms, double speech_noise_thres) {
    pf::Decision* d = reinterpret_cast<pf::Decision*>(dh);
    if (!d) return;
    d->max_end_sil_ms = max_end_sil_ms;
    d->speech_noise_thres = speech_noise_thres;
}
int pf_vad_decision_state(const pf_vad_decision* dh) {
    const pf::Decision* d = reinterpret_cast<const pf::Decision*>(dh);
    return d ? d->state : -1;
}
int pf_vad_decision_push(pf_vad_decision* dh, const float* sil_scores, const float* decibels, int32_t n_frames,
                         int32_t is_final, int32_t streaming_events, int32_t* segments_out, int32_t capacity) {
    pf::Decision* d = reinterpret_cast<pf::Decision*>(dh);
    PF_REQUIRE(d && (n_frames <= 0 || (sil_scores && decibels)) && (capacity <= 0 || segments_out), "vad_decision_push: null");
    const int m = d->push(sil_scores, decibels, n_frames, is_final != 0, streaming_events != 0, segments_out, capacity);
    if (m < 0) { pf::set_error("vad_decision: " + d->error); return -1; }
    return m;
}

}  // extern "C"
