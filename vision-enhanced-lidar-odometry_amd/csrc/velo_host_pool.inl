// velo_host_pool.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  Resident host threads for the batch entry points (WorkerPool).
// =====================================================================================================================
// Resident host threads for the batch entry points: a step of 8 pairs used to create and join 3 (lock-step groups) or 8 (one per
// context) std::threads inside the timed region, every step.  The workers are created on first need and then wait on a condition
// variable between calls; a call hands out task indices 1..n-1 and runs task 0 itself.  One call at a time owns the pool -- a
// second caller that arrives meanwhile (another host thread driving another device) spawns its own threads as before.
class WorkerPool {
public:
    static WorkerPool& instance() { static WorkerPool p; return p; }
    template <typename F>
    void run(int n, F&& fn) {
        if (n <= 1) { if (n == 1) fn(0); return; }
        std::unique_lock<std::mutex> owner(owner_, std::try_to_lock);
        if (!owner.owns_lock()) {                                   // pool busy: plain threads for this call
            std::vector<std::thread> th;
            for (int i = 1; i < n; i++) th.emplace_back([&fn, i]() { fn(i); });
            fn(0);
            for (auto& t : th) t.join();
            return;
        }
        std::function<void(int)> f = [&fn](int i) { fn(i); };
        {
            std::lock_guard<std::mutex> lk(m_);
            while ((int)workers_.size() < n - 1) workers_.emplace_back([this]() { loop(); });
            fn_ = &f; next_ = 1; n_ = n; pending_ = n - 1; generation_++;
        }
        wake_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }
private:
    void loop() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            wake_.wait(lk, [&]() { return stop_ || (generation_ != seen && next_ < n_); });
            if (stop_) return;
            while (next_ < n_) {
                const int i = next_++;
                const std::function<void(int)>* f = fn_;
                lk.unlock();
                (*f)(i);
                lk.lock();
                if (--pending_ == 0) done_.notify_all();
            }
            seen = generation_;
        }
    }
    std::mutex owner_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, n_ = 0, pending_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
};

static int associate_target_sharded(velo_ctx* c, const double x[6], int iter, bool want_aux);
static int launch_merge(velo_ctx* c, const PartialRec* tables, int world, int stride, int iter, bool want_aux);
