// oracle/ref_shim/boost/bind.hpp -- TEST INFRASTRUCTURE.  Boost is not installed here.  boost::bind as the reference uses it: member functions
// / member data / free functions with bound leading arguments and the global placeholders _1.._4, plus `bind(...) < bind(...)`
// (Reprojector.cpp: sorting by a member).  Built on std::bind.
#pragma once
#include <functional>
#include <utility>

namespace boost {
namespace _bi {
template <typename F> struct bind_t {
    F f;
    template <typename... A> auto operator()(A&&... a) -> decltype(f(std::forward<A>(a)...)) { return f(std::forward<A>(a)...); }
    template <typename... A> auto operator()(A&&... a) const -> decltype(f(std::forward<A>(a)...)) { return f(std::forward<A>(a)...); }
};
template <typename L, typename R> struct less_t {
    L l; R r;
    template <typename... A> bool operator()(A&&... a) { return l(a...) < r(a...); }
};
template <typename L, typename R> less_t<bind_t<L>, bind_t<R> > operator<(const bind_t<L>& l, const bind_t<R>& r) { return less_t<bind_t<L>, bind_t<R> >{l, r}; }
}  // namespace _bi
template <typename F, typename... A>
auto bind(F&& f, A&&... a) -> _bi::bind_t<decltype(std::bind(std::forward<F>(f), std::forward<A>(a)...))> {
    return _bi::bind_t<decltype(std::bind(std::forward<F>(f), std::forward<A>(a)...))>{std::bind(std::forward<F>(f), std::forward<A>(a)...)};
}
}  // namespace boost
using std::placeholders::_1;
using std::placeholders::_2;
using std::placeholders::_3;
using std::placeholders::_4;
