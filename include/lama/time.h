// lama/time.h -- lama::Duration / lama::Time with the reference's interface (include/lama/time.h:44-250): a nanosecond duration
// and a wall-clock time point built on it.  Consumers (iris_lama_ros) use them for stamps and throttling; nothing on the device
// path depends on them.  Own implementation on std::chrono.
#pragma once

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
        if (left <= Duration(0.0)) {                     // overran (or the clock jumped): re-anchor when a full cycle was lost
            if (actual_cycle > cycle || t < start - cycle) start = t;
            return;
        }
        left.sleep();
    }
    void reset() { start = Time::now(); }
    Duration cycleTime() const { return actual_cycle; }
};

} // namespace lama
