// fwgpu_msgq.h — the two lock-free rings between the control side and the audio thread of one ctx.
//
// The reference gives every node its own channel: the gain of a VolumeNode / SamplerNode is an Arc<AtomicF32> any
// thread may store to (nodes/volume.rs:10,28-34, nodes/sampler.rs:49,171-177), a SamplerNode talks to its processor
// through an rtrb SPSC ring of CHANNEL_CAPACITY messages (sampler.rs:14,205-208) and gets swapped-out samples back
// through a second ring (ProcessorToNodeMsg::ReturnSample, sampler.rs:339-343,563-571).  Here one device call covers
// many blocks and every message carries the block it takes effect at, so both directions are ONE ring per ctx:
//   MsgRing  control -> audio: bounded multi-producer / single-consumer queue of Cmd.  Setters (fwgpu_node_set_param,
//            fwgpu_sampler_*) push from any thread; the audio thread pops everything at the start of a process call.
//            Neither side locks, allocates or waits for the other (a full ring is FWGPU_ERR_QUEUE_FULL).
//   RetRing  audio -> control: single-producer / single-consumer queue of (node, sample id, completion ticket).
// Bounded MPMC scheme after D. Vyukov: every cell carries a sequence number that says whose turn it is.
#pragma once
#include <stdint.h>

#include <atomic>
#include <new>

#include "fwgpu_types.h"

namespace fwgpu {

class MsgRing {
  public:
    MsgRing() = default;
    MsgRing(const MsgRing&) = delete;
    MsgRing& operator=(const MsgRing&) = delete;
    ~MsgRing() { delete[] cells_; }
    bool init(uint32_t capacity_pow2) {  // control thread, before the ctx is shared
        cells_ = new (std::nothrow) Cell[capacity_pow2];
        if (!cells_) return false;
        mask_ = capacity_pow2 - 1;
        for (uint64_t i = 0; i <= mask_; ++i) cells_[i].seq.store(i, std::memory_order_relaxed);
        tail_.store(0, std::memory_order_relaxed);
        head_ = 0;
        return true;
    }
    // any thread.  false = full.
    bool push(const Cmd& m) {
        uint64_t pos = tail_.load(std::memory_order_relaxed);
        Cell* cell;
        for (;;) {
            cell = &cells_[pos & mask_];
            const uint64_t seq = cell->seq.load(std::memory_order_acquire);
            const int64_t dif = (int64_t)seq - (int64_t)pos;
            if (dif == 0) {
                if (tail_.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) break;
            } else if (dif < 0) {
                return false;  // the consumer has not freed this cell yet: full
            } else {
                pos = tail_.load(std::memory_order_relaxed);
            }
        }
        cell->cmd = m;
        cell->seq.store(pos + 1, std::memory_order_release);
        return true;
    }
    // the consumer (one thread at a time).  false = empty (or the next producer has reserved its cell but not
    // published it yet: it will be seen by the next drain — exactly like a message sent a moment later)
    bool pop(Cmd& out) {
        Cell* cell = &cells_[head_ & mask_];
        const uint64_t seq = cell->seq.load(std::memory_order_acquire);
        if (seq != head_ + 1) return false;
        out = cell->cmd;
        cell->seq.store(head_ + mask_ + 1, std::memory_order_release);
        head_++;
        return true;
    }

  private:
    struct Cell {
        std::atomic<uint64_t> seq;
        Cmd cmd;
    };
    Cell* cells_ = nullptr;
    uint64_t mask_ = 0;
    alignas(64) std::atomic<uint64_t> tail_{0};
    alignas(64) uint64_t head_ = 0;  // consumer only
};

constexpr int64_t RET_SILENT = INT64_MIN;  // RetItem::node of a return that only moves the reference count (a dropped sampler's
                                           // sample, a SetSample message that never reached its node): nothing is reported
struct RetItem {
    int64_t node;     // the sampler node that let go of the sample
    int sample;       // sample id (fwgpu_sample_create)
    uint32_t ticket;  // index of the process call (among those that returned something): its completion event
};

class RetRing {  // SPSC: the audio thread pushes, the control side pops
  public:
    RetRing() = default;
    RetRing(const RetRing&) = delete;
    RetRing& operator=(const RetRing&) = delete;
    ~RetRing() { delete[] items_; }
    bool init(uint32_t capacity_pow2) {
        items_ = new (std::nothrow) RetItem[capacity_pow2];
        mask_ = capacity_pow2 - 1;
        return items_ != nullptr;
    }
    bool push(const RetItem& it) {
        const uint64_t t = tail_.load(std::memory_order_relaxed);
        if (t - head_.load(std::memory_order_acquire) > mask_) return false;
        items_[t & mask_] = it;
        tail_.store(t + 1, std::memory_order_release);
        return true;
    }
    // two-phase push (producer only): items are written behind the published tail and become visible together — a process call
    // stages its returns while it retires messages and publishes them only AFTER their completion event has been recorded
    bool stage(const RetItem& it) {
        const uint64_t t = tail_.load(std::memory_order_relaxed) + staged_;
        if (t - head_.load(std::memory_order_acquire) > mask_) return false;
        items_[t & mask_] = it;
        staged_++;
        return true;
    }
    void publish() {
        if (!staged_) return;
        tail_.store(tail_.load(std::memory_order_relaxed) + staged_, std::memory_order_release);
        staged_ = 0;
    }
    bool peek(RetItem& out) const {
        const uint64_t h = head_.load(std::memory_order_relaxed);
        if (h == tail_.load(std::memory_order_acquire)) return false;
        out = items_[h & mask_];
        return true;
    }
    void pop() { head_.store(head_.load(std::memory_order_relaxed) + 1, std::memory_order_release); }

  private:
    RetItem* items_ = nullptr;
    uint64_t mask_ = 0;
    uint64_t staged_ = 0;  // producer side only
    alignas(64) std::atomic<uint64_t> tail_{0};
    alignas(64) std::atomic<uint64_t> head_{0};
};

}  // namespace fwgpu
