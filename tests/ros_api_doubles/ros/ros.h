// Test double (see ../README.md): the roscpp declarations GpPredictor's ROS configuration uses, recording
// instead of talking to a ROS master.
#pragma once
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>
namespace ros {
struct Bus {   // what the class did, and the canned behaviour of the world around it
  std::map<std::string, std::pair<int, std::function<void(const void *)>>> subscriptions;   // topic -> (queue, callback)
  std::map<std::string, int> advertised;                                                    // topic -> queue
  std::vector<std::string> service_clients;
  std::vector<std::pair<std::string, double>> published;                                    // (topic, data)
  std::function<bool(const std::string &, void *)> service;                                 // fills the response
  std::function<double()> clock;
  std::map<std::string, double> params;
  std::string node_name;
  int spins = 0;
  std::vector<std::string> log;
};
inline Bus &bus() { static Bus b; return b; }
struct Time {
  double t = 0;
  double toSec() const { return t; }
  static Time now() { Time x; x.t = bus().clock ? bus().clock() : 0.0; return x; }
};
struct Subscriber {};
struct Publisher {
  std::string topic;
  template <class M> void publish(const M &m) const { bus().published.emplace_back(topic, (double)m.data); }
};
struct ServiceClient {
  std::string name;
  template <class S> bool call(S &srv) { return bus().service && bus().service(name, &srv); }
};
struct NodeHandle {
  std::string ns;
  NodeHandle(const std::string &n = "") : ns(n) {}
  template <class M, class T>
  Subscriber subscribe(const std::string &topic, int queue, void (T::*fp)(const std::shared_ptr<const M> &), T *obj) {
    bus().subscriptions[topic] = {queue, [obj, fp](const void *p) { (obj->*fp)(*static_cast<const std::shared_ptr<const M> *>(p)); }};
    return Subscriber();
  }
  template <class S> ServiceClient serviceClient(const std::string &name) {
    bus().service_clients.push_back(name);
    ServiceClient c; c.name = name; return c;
  }
  template <class M> Publisher advertise(const std::string &topic, int queue) {
    bus().advertised[topic] = queue;
    Publisher p; p.topic = topic; return p;
  }
  bool getParam(const std::string &key, double &out) const {
    auto it = bus().params.find(key);
    if (it == bus().params.end()) return false;
    out = it->second;
    return true;
  }
};
inline void init(int &, char **, const std::string &name) { bus().node_name = name; }
inline void spin() { ++bus().spins; }
}  // namespace ros
