// hostpool.hpp — a handful of persistent host threads for the per-proof host work of small batches (the verifier's transcript hashing,
// csrc/rp.hip).  Starting a std::thread per proof costs ~30 us each, more than the 36 us of hashing it would carry; the pool's workers
// sleep on a condition variable between calls and the caller takes its share of the items.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace bppp {

class HostPool {
 public:
  explicit HostPool(unsigned workers) {
    for (unsigned i = 0; i < workers; i++) th_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    start_.notify_all();
    for (auto &t : th_) t.join();
  }
  HostPool(const HostPool &) = delete;
  HostPool &operator=(const HostPool &) = delete;
  // f(i) for every i < count, on the workers and on the calling thread; returns when all are done.  One call at a time.
  void run(size_t count, const std::function<void(size_t)> &f) {
    if (count <= 1 || th_.empty()) { for (size_t i = 0; i < count; i++) f(i); return; }
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f; count_ = count; next_.store(0); busy_ = th_.size(); gen_++;
    }
    start_.notify_all();
    for (size_t i; (i = next_.fetch_add(1)) < count;) f(i);
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return busy_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop() {
    size_t seen = 0;
    for (;;) {
      const std::function<void(size_t)> *f;
      size_t count;
      {
        std::unique_lock<std::mutex> g(m_);
        start_.wait(g, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_; f = job_; count = count_;
      }
      for (size_t i; (i = next_.fetch_add(1)) < count;) (*f)(i);
      {
        std::lock_guard<std::mutex> g(m_);
        if (--busy_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable start_, done_;
  const std::function<void(size_t)> *job_ = nullptr;
  size_t count_ = 0, gen_ = 0, busy_ = 0;
  std::atomic<size_t> next_{0};
  bool stop_ = false;
};

}  // namespace bppp
