// ffh_abi_guard.hpp -- FFH_CATCH (part of the ONE translation unit ffh_api.hip; also included by tests/pipe_emul_main.cpp)
#pragma once
#include <new>
#include <string>

// No C++ exception crosses the C ABI (a JVM that loads the library through JNI would be taken down by one): the entry points that allocate host
// memory are function-try-blocks that end in FFH_CATCH -- std::bad_alloc becomes FFH_E_NOMEM, anything else FFH_E_STATE, the message goes where
// the entry point's other errors go.  Walked on the CPU: tests/mock_hip/fault_main.c with the n-th `operator new` throwing (round 6).
static inline void ffh_note_exception(std::string *err, const char *what) noexcept { try { if (err) *err = what; } catch (...) {} }
#define FFH_CATCH(errp)                                                                                                   \
    catch (const std::bad_alloc &) { ffh_note_exception((errp), "out of host memory"); return FFH_E_NOMEM; }               \
    catch (const std::exception &e_) { ffh_note_exception((errp), e_.what()); return FFH_E_STATE; }                         \
    catch (...) { ffh_note_exception((errp), "unknown C++ exception"); return FFH_E_STATE; }

