/* Declared test double of the part of roscpp that ros/lightloam_*_node.cpp use -- NOT ROS.  This image has no ROS; the
 * double lets the node sources be compiled as they are and their callbacks driven from a test: advertise() records the
 * topic, publish() keeps the last message per topic, subscribe() keeps the callback, spin() returns at once. */
#pragma once
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace ros {

struct Time {
    unsigned sec = 0, nsec = 0;
    bool operator==(const Time &o) const { return sec == o.sec && nsec == o.nsec; }
    double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
    Time &fromSec(double t) { sec = (unsigned)t; nsec = (unsigned)((t - (double)sec) * 1e9 + 0.5); return *this; }
};
struct Rate { explicit Rate(double) {} void sleep() {} };

struct Double {                                   /* the state a test inspects */
    std::map<std::string, int> advertised;        /* topic -> queue size */
    std::map<std::string, int> subscribed;
    std::map<std::string, std::shared_ptr<void>> last;        /* topic -> last published message */
    std::map<std::string, int> published;                     /* topic -> count */
    std::map<std::string, double> params_d; std::map<std::string, int> params_i; std::map<std::string, std::string> params_s;
    std::function<void(const std::shared_ptr<const void> &)> callback;        /* of the last subscription */
    std::map<std::string, std::function<void(const std::shared_ptr<const void> &)>> callbacks;   /* topic -> callback */
    std::function<bool()> on_spin;                /* spinOnce(): the test delivers messages here; false ends ros::ok() */
    bool running = true;
    std::mutex mu;                                 /* publish() may come from a node's worker thread */
    int count(const std::string &topic) { std::lock_guard<std::mutex> l(mu); return published.count(topic) ? published[topic] : 0; }
    std::shared_ptr<void> latest(const std::string &topic) { std::lock_guard<std::mutex> l(mu); return last.count(topic) ? last[topic] : nullptr; }
    int warnings = 0, errors = 0;
    static Double &get() { static Double d; return d; }
};

inline void init(int &, char **, const std::string &) {}
inline void spin()                                 /* the test's on_spin delivers messages until it returns false */
{
    auto &D = Double::get();
    while (D.on_spin && D.on_spin()) {}
    D.running = false;
}
inline void spinOnce() { auto &D = Double::get(); if (D.on_spin) D.running = D.on_spin(); else D.running = false; }
inline bool ok() { return Double::get().running; }

class Publisher {
public:
    std::string topic;
    template <class M> void publish(const M &m) const {
        auto &D = Double::get();
        std::lock_guard<std::mutex> l(D.mu);
        D.last[topic] = std::make_shared<M>(m);
        D.published[topic]++;
    }
};
class Subscriber {};

class NodeHandle {
public:
    void param(const std::string &name, std::string &out, const std::string &dflt) const {
        auto &D = Double::get();
        out = D.params_s.count(name) ? D.params_s[name] : dflt;
    }
    template <class T> void param(const std::string &name, T &out, const T &dflt) const {
        auto &D = Double::get();
        if (D.params_d.count(name)) out = (T)D.params_d[name];
        else if (D.params_i.count(name)) out = (T)D.params_i[name];
        else out = dflt;
    }
    template <class M> Publisher advertise(const std::string &topic, int queue) {
        Double::get().advertised[topic] = queue;
        Publisher p; p.topic = topic; return p;
    }
    template <class M> Subscriber subscribe(const std::string &topic, int queue, void (*cb)(const std::shared_ptr<const M> &)) {
        Double::get().subscribed[topic] = queue;
        Double::get().callback = [cb](const std::shared_ptr<const void> &m) { cb(std::static_pointer_cast<const M>(m)); };
        Double::get().callbacks[topic] = Double::get().callback;
        return Subscriber();
    }
};

}  // namespace ros

#define ROS_WARN(...) do { ros::Double::get().warnings++; } while (0)
#define ROS_BREAK() do { std::fprintf(stderr, "ROS_BREAK\n"); std::abort(); } while (0)
#define ROS_ERROR(...) do { ros::Double::get().errors++; std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
