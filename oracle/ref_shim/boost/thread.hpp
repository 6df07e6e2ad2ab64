// ORACLE-SIDE TEST INFRASTRUCTURE — stand-in for the subset of Boost.Thread / Boost.Bind / Boost.Function the reference's worker
// pool (util/IndexThreadReduce.h) uses, mapped one-to-one onto the C++11 standard library.  Boost is not installed in this image.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
namespace boost
{
typedef std::thread thread;
typedef std::mutex mutex;
typedef std::condition_variable condition_variable;
template<class M> using unique_lock = std::unique_lock<M>;
template<class M> using lock_guard = std::lock_guard<M>;
template<class Sig> using function = std::function<Sig>;
using std::bind;
using std::ref;
namespace placeholders
{
using std::placeholders::_1;
using std::placeholders::_2;
using std::placeholders::_3;
using std::placeholders::_4;
using std::placeholders::_5;
using std::placeholders::_6;
}
namespace this_thread { using std::this_thread::get_id; using std::this_thread::yield; }
}
