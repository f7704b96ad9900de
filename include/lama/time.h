// lama/time.h -- lama::Duration / lama::Time with the reference's interface (include/lama/time.h:44-250): a nanosecond duration
// and a wall-clock time point built on it.  Consumers (iris_lama_ros) use them for stamps and throttling; nothing on the device
// path depends on them.  Own implementation on std::chrono.
#pragma once

#include <queue>
#include <chrono>
#include <cstdint>
#include <thread>

#include "types.h"

namespace lama {

struct Duration {
    std::chrono::nanoseconds ns{0};

    Duration() = default;
    explicit Duration(double seconds) : ns((int64_t)(seconds * 1.e9)) {}
    Duration(const std::chrono::nanoseconds& nano) : ns(nano) {}

    bool isZero() const { return ns.count() == 0; }
    void sleep() const { std::this_thread::sleep_for(ns); }
    double toSec() const { return (double)ns.count() / 1.e9; }
    int64_t toNSec() const { return ns.count(); }

    bool operator==(const Duration& o) const { return ns == o.ns; }
    bool operator!=(const Duration& o) const { return ns != o.ns; }
    bool operator<(const Duration& o) const { return ns < o.ns; }
    bool operator>(const Duration& o) const { return ns > o.ns; }
    bool operator<=(const Duration& o) const { return ns <= o.ns; }
    bool operator>=(const Duration& o) const { return ns >= o.ns; }

    Duration operator+(const Duration& o) const { return Duration(ns + o.ns); }
    Duration operator-(const Duration& o) const { return Duration(ns - o.ns); }
    Duration operator*(double k) const { return Duration(toSec() * k); }
    Duration operator-() const { return Duration(-ns); }
    Duration& operator+=(const Duration& o) { ns += o.ns; return *this; }
    Duration& operator-=(const Duration& o) { ns -= o.ns; return *this; }
};

struct Time {
    Duration duration;

    Time() = default;
    Time(const Duration& since_epoch) : duration(since_epoch) {}
    explicit Time(double seconds) : duration(seconds) {}

    static Time now() { return Time(Duration(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()))); }
    static void sleepUntil(const Time& end) { const Duration d = end - now(); if (d > Duration(0.0)) d.sleep(); }

    bool isZero() const { return duration.isZero(); }
    double toSec() const { return duration.toSec(); }
    int64_t toNSec() const { return duration.toNSec(); }

    bool operator==(const Time& o) const { return duration == o.duration; }
    bool operator!=(const Time& o) const { return duration != o.duration; }
    bool operator<(const Time& o) const { return duration < o.duration; }
    bool operator>(const Time& o) const { return duration > o.duration; }
    bool operator<=(const Time& o) const { return duration <= o.duration; }
    bool operator>=(const Time& o) const { return duration >= o.duration; }

    Duration operator-(const Time& o) const { return duration - o.duration; }
    Time operator+(const Duration& d) const { return Time(duration + d); }
    Time operator-(const Duration& d) const { return Time(duration - d); }
    Time& operator+=(const Duration& d) { duration += d; return *this; }
    Time& operator-=(const Duration& d) { duration -= d; return *this; }
};

// stopwatch on Time::now() (include/lama/time.h: Timer)
struct Timer {
    Time time_point;
    explicit Timer(bool immediately = false) { if (immediately) start(); }
    void start() { time_point = Time::now(); }
    void reset() { start(); }
    Duration elapsed() const { return Time::now() - time_point; }
};

// fixed-rate loop helper (include/lama/time.h: Rate): sleep() waits for what is left of the cycle; `actual_cycle` is the
// measured length of the cycle that just ended
struct Rate {
    Time start;
    Duration cycle, actual_cycle;
    explicit Rate(double frequency) : start(Time::now()), cycle(1.0 / frequency), actual_cycle(0.0) {}
    void sleep()
    {
        const Time t = Time::now();
        const Time end = (t < start ? t : start) + cycle;
        const Duration left = end - t;
        actual_cycle = t - start;
        start = end;
        if (left <= Duration(0.0)) {                     // overran: keep the phase and catch up, unless more than a full cycle
            if (t > end + cycle) start = t;              // was lost (or the clock jumped) -- only then re-anchor at `now`
            return;
        }
        left.sleep();
    }
    void reset() { start = Time::now(); }
    Duration cycleTime() const { return actual_cycle; }
};

// rate of the last `window` events (include/lama/time.h: EventFrequency): event() stamps the stopwatch's elapsed time, event(t)
// a caller-supplied (simulated) time; getFrequency() = (events - 1) / (newest - oldest), 0 with fewer than two events
struct EventFrequency {
    const uint32_t window;
    Timer timer;
    std::queue<Duration> queue;
    explicit EventFrequency(uint32_t window_size = 30) : window(window_size) { timer.start(); }
    void reset() { timer.reset(); queue = std::queue<Duration>(); }
    void event() { push(timer.elapsed()); }
    void event(const double timestamp) { push(Duration(timestamp)); }
    double getFrequency() const
    {
        if (queue.size() < 2) return 0.0;
        return (double)(queue.size() - 1) / (queue.back() - queue.front()).toSec();
    }
private:
    void push(const Duration& d) { queue.push(d); if (queue.size() > window) queue.pop(); }
};

} // namespace lama
