// gtop_guard.h — the exception barrier of the C-ABI (include/gtop.h: "no exceptions cross this boundary").
//
// The entry points are C++ inside (std::string error texts, std::vector bookkeeping), so an allocation failure would
// otherwise unwind through an `extern "C"` frame into a C, NLopt or ctypes caller — undefined behaviour.  Every entry
// point that can allocate is a function-try-block ending in one of these macros: whatever is thrown becomes
// GTOP_ERR_INTERNAL (HUGE_VAL for the two nlopt_func-shaped callbacks) and, where there is an object to hold it, an
// error text.  `note` is a callable (object, const char *) that must not throw.
#ifndef GTOP_GUARD_H_
#define GTOP_GUARD_H_

#include <cmath>
#include <exception>
#include <new>

#define GTOP_CATCH_WITH(note, obj, ret)                                                   \
  catch (const std::bad_alloc &) { note(obj, "out of memory (std::bad_alloc)"); return ret; } \
  catch (const std::exception &e_) { note(obj, e_.what()); return ret; }                  \
  catch (...) { note(obj, "unknown C++ exception"); return ret; }

#endif  // GTOP_GUARD_H_
