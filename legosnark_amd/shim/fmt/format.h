// Minimal stand-in for the {fmt} calls LegoSNARK makes: fmt::print / fmt::format with bare
// "{}" placeholders only (benchmark.cc:5, benchmark.h:108,134,144, lipmaa.h:70,118-119,
// lipmaa.cc:179, subspace.h:63-64).  Not a general formatter.
#pragma once
#include <cstdio>
#include <sstream>
#include <string>

namespace fmt {
namespace detail {
inline void format_to(std::ostringstream &os, const char *f) { os << f; }
template <class T, class... Rest>
void format_to(std::ostringstream &os, const char *f, const T &v, const Rest &...rest) {
    for (; *f; ++f) {
        if (f[0] == '{' && f[1] == '}') { os << v; format_to(os, f + 2, rest...); return; }
        os << *f;
    }
}
}  // namespace detail
template <class... Args>
std::string format(const char *f, const Args &...args) {
    std::ostringstream os;
    detail::format_to(os, f, args...);
    return os.str();
}
template <class... Args>
std::string format(const std::string &f, const Args &...args) { return format(f.c_str(), args...); }
template <class... Args>
void print(const char *f, const Args &...args) { std::fputs(format(f, args...).c_str(), stdout); }
}  // namespace fmt
