// oracle/ref_shim/boost/thread.hpp -- TEST INFRASTRUCTURE.  Boost is not installed here.  The reference uses boost::thread / mutex /
// condition_variable / unique_lock / function / bind (util/IndexThreadReduce.h, FullSystem.h, Reprojector.cpp); they map one to one onto the
// C++11 standard library, which is what this header does so that those files compile unmodified into oracle/_ref/libref.so.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "boost/bind.hpp"

namespace boost {
using std::thread;
using std::mutex;
using std::condition_variable;
using std::unique_lock;
using std::lock_guard;
template <typename Sig> class function : public std::function<Sig> {
public:
    using std::function<Sig>::function;
    bool operator!=(int) const { return static_cast<bool>(*this); }   // `assert(callPerIndex != 0)` in IndexThreadReduce.h
    bool operator==(int) const { return !static_cast<bool>(*this); }
};
namespace this_thread { using namespace std::this_thread; }
}  // namespace boost
