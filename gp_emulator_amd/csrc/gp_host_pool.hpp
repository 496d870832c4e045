// Persistent helper threads for the host side of the host-pointer predict path
// (gp_abi.hip, predict_host): copying slabs of the caller's pageable arrays into and out of
// the pinned staging buffers, converting float64 <-> float32 on the way.
//
// One pool per gp_ctx (a context is driven by one thread at a time), created on the first
// large call and kept until gp_ctx_destroy: nothing is spawned per call or per slab.  The
// calling thread takes part in every job, so a pool of n threads has n - 1 workers.  Tasks
// of a job are claimed with one atomic add each (dynamic balance: a helper that is
// descheduled only delays its own task).  Idle workers spin for about 0.2 ms before they
// block, because inside a pipelined call, or a loop of small calls, the next job arrives that soon.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include <pthread.h>
#include <sched.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace gph {

inline void cpu_relax() {
#if defined(__x86_64__)
  _mm_pause();
#endif
}

class ThreadPool {
 public:
  // cpus != nullptr: the workers may only run on those cpus (the caller's thread is not touched)
  explicit ThreadPool(int n_threads, const cpu_set_t* cpus = nullptr) {
    const int workers = n_threads > 1 ? n_threads - 1 : 0;
    for (int i = 0; i < workers; ++i) {
      threads_.emplace_back([this] { worker(); });
      if (cpus) (void)pthread_setaffinity_np(threads_.back().native_handle(), sizeof(cpu_set_t), cpus);
    }
  }
  ~ThreadPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  ThreadPool(const ThreadPool&) = delete;
  ThreadPool& operator=(const ThreadPool&) = delete;

  int size() const { return (int)threads_.size() + 1; }

  // f() on one of the workers (on the calling thread when the pool has none); returns when done
  template <typename F>
  void run_on_worker(F&& f) {
    if (threads_.empty()) { f(); return; }
    Job job;
    auto body = [&](int) { f(); };
    job.call = [](void* ctx, int t) { (*static_cast<decltype(body)*>(ctx))(t); };
    job.ctx = &body;
    job.n = 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &job;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    // the caller does NOT take part: wait for a worker to claim and finish the one task
    while (job.done.load(std::memory_order_acquire) < 1) std::this_thread::yield();
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = nullptr;
    }
    while (job.attached.load(std::memory_order_acquire) != 0) cpu_relax();
  }

  // fn(task) for every task in [0, n_tasks); returns when all of them have finished.
  template <typename F>
  void run(int n_tasks, F&& fn) {
    if (n_tasks <= 0) return;
    if (threads_.empty() || n_tasks == 1) {
      for (int t = 0; t < n_tasks; ++t) fn(t);
      return;
    }
    Job job;
    job.call = [](void* ctx, int t) { (*static_cast<typename std::remove_reference<F>::type*>(ctx))(t); };
    job.ctx = &fn;
    job.n = n_tasks;
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &job;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    work(job);
    // all tasks claimed; wait for the helpers still inside one, and for every helper that
    // attached to this job to let go of it (it lives on this stack frame)
    while (job.done.load(std::memory_order_acquire) < n_tasks) cpu_relax();
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = nullptr;
    }
    while (job.attached.load(std::memory_order_acquire) != 0) cpu_relax();
  }

 private:
  struct Job {
    void (*call)(void*, int) = nullptr;
    void* ctx = nullptr;
    int n = 0;
    std::atomic<int> next{0}, done{0}, attached{0};
  };

  static void work(Job& job) {
    for (;;) {
      const int t = job.next.fetch_add(1, std::memory_order_relaxed);
      if (t >= job.n) break;
      job.call(job.ctx, t);
      job.done.fetch_add(1, std::memory_order_release);
    }
  }

  void worker() {
    unsigned long long seen = 0;
    for (;;) {
      // short spin, then block
      // (~0.2 ms: inside a pipelined call the next job arrives within tens of microseconds, and a caller that asks
      // for one state vector at a time -- whose per-call content check takes one helper -- comes back every ~0.1 ms)
      for (int i = 0; i < 16000 && gen_.load(std::memory_order_acquire) == seen; ++i) cpu_relax();
      Job* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        seen = gen_.load(std::memory_order_acquire);
        if (stop_) return;
        job = job_;
        if (job) job->attached.fetch_add(1, std::memory_order_acq_rel);
      }
      if (job) {
        work(*job);
        job->attached.fetch_sub(1, std::memory_order_acq_rel);
      }
    }
  }

  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::atomic<unsigned long long> gen_{0};
  Job* job_ = nullptr;
  bool stop_ = false;
};

}  // namespace gph
