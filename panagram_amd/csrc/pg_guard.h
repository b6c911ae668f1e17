// pg_guard.h — exception firewall of the C-ABI (include/panagram_hip.h: "no C++ exception crosses the ABI").
// Every extern "C" body that can reach an allocation, a std::thread or a container runs inside
// PG_API_BEGIN / PG_API_END: a std::bad_alloc becomes PG_E_CAPACITY, anything else PG_E_INVALID, the text goes to
// pg_last_error().  The reference's convention at this seam is an error return, never an abort of the interpreter
// (panagram/index.py:850-853 re-raises import errors with a hint; KMC's calls return bool).
#pragma once
#include <exception>
#include <new>

int pg_set_error(int code, const char *msg);  // pg_api.hip: the library's thread-local error slot

namespace pg {
template <class F>
static inline int guarded(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try {
            return pg_set_error(-4 /* PG_E_CAPACITY */, "out of host memory");
        } catch (...) {
            return -4;
        }
    } catch (const std::exception &e) {
        try {
            return pg_set_error(-1 /* PG_E_INVALID */, e.what());
        } catch (...) {
            return -1;
        }
    } catch (...) {
        try {
            return pg_set_error(-1 /* PG_E_INVALID */, "unknown C++ exception");
        } catch (...) {
            return -1;
        }
    }
}
}  // namespace pg
#define PG_API_BEGIN return pg::guarded([&]() -> int {
#define PG_API_END });
