// slot_queue.h -- a fixed set of buffers ("slots") cycling between one producer and one consumer thread:
// the producer acquire()s a slot whose previous content has been consumed, fills it, submit()s a job that refers to
// it; the consumer thread runs the jobs in order and returns each slot.  Used by `dashing2 cmp` (device batch i+1
// under the emit of batch i) -- the overlap the reference gets from its writer thread, src/emitrect.cpp:159-197.
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>

namespace d2h {

template <class Job>
class SlotQueue {
    struct Item { int slot; Job job; };
    std::function<void(const Job &)> consume_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Item> q_;
    std::deque<int> free_;
    bool closing_ = false;
    std::thread th_;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
public:
    double t_busy = 0;                               // seconds the consumer spent inside consume(); read after finish()
    SlotQueue(int nslots, std::function<void(const Job &)> consume) : consume_(std::move(consume)) {
        for (int i = 0; i < nslots; ++i) free_.push_back(i);
        th_ = std::thread([this] {
            for (;;) {
                Item it;
                {
                    std::unique_lock<std::mutex> lk(m_);
                    cv_.wait(lk, [&] { return closing_ || !q_.empty(); });
                    if (q_.empty()) return;
                    it = std::move(q_.front());
                    q_.pop_front();
                }
                const double t0 = now();
                consume_(it.job);
                const double dt = now() - t0;
                { std::lock_guard<std::mutex> lk(m_); free_.push_back(it.slot); t_busy += dt; }
                cv_.notify_all();
            }
        });
    }
    SlotQueue(const SlotQueue &) = delete;
    SlotQueue &operator=(const SlotQueue &) = delete;
    int acquire() {                                   // blocks until a slot is free
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !free_.empty(); });
        const int s = free_.front();
        free_.pop_front();
        return s;
    }
    void submit(int slot, Job job) {
        { std::lock_guard<std::mutex> lk(m_); q_.push_back(Item{slot, std::move(job)}); }
        cv_.notify_all();
    }
    void finish() {                                   // drains the queue, joins the consumer
        { std::lock_guard<std::mutex> lk(m_); closing_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    ~SlotQueue() { finish(); }
};

}  // namespace d2h
