// abi_barrier.hpp -- the exception barrier of the C ABI (include/sicp.h: "no exceptions cross the ABI").
//
// Every extern "C" entry point that returns a status runs its body inside abi_guard: the host side uses
// std::vector / std::string / std::deque / std::thread, so std::bad_alloc and std::system_error are possible in
// most of them, and an exception that unwinds through an extern "C" frame into a C caller (or ctypes, cgo, JNI)
// is undefined behaviour.  bad_alloc -> SICP_ERR_OUT_OF_MEMORY, anything else -> SICP_ERR_INTERNAL; `note`
// receives a description for sicp_last_error (it must not throw; abi_guard swallows it if it does).
// Self-contained on purpose: tests/test_abi.py compiles it alone with g++ and checks the mapping.
#ifndef SICP_ABI_BARRIER_HPP_
#define SICP_ABI_BARRIER_HPP_

#include <exception>
#include <new>

#include "sicp.h"

namespace sicp {
namespace host {

template <class Body, class Note>
inline int abi_guard_note(Body&& body, Note&& note) noexcept {
  int status = SICP_ERR_INTERNAL;
  const char* what = nullptr;
  try {
    return body();
  } catch (const std::bad_alloc&) {
    status = SICP_ERR_OUT_OF_MEMORY;
    what = "out of host memory (std::bad_alloc)";
  } catch (const std::exception& e) {
    try { note(e.what()); } catch (...) {}
    return SICP_ERR_INTERNAL;
  } catch (...) {
    what = "unknown C++ exception";
  }
  try { note(what); } catch (...) {}
  return status;
}

template <class Body>
inline int abi_guard(Body&& body) noexcept {
  return abi_guard_note(static_cast<Body&&>(body), [](const char*) {});
}

}  // namespace host
}  // namespace sicp
#endif
