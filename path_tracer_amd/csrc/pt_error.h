// pt_error.h — the thread-local error message behind pt_last_error(), shared by the translation
// units of the library (internal: not part of the ABI, hidden from the export list).
#pragma once

// Stores the message, returns `code`.
__attribute__((visibility("hidden"), format(printf, 2, 3))) int pt_fail(int code, const char* fmt, ...);
// Every extern "C" entry point calls this first, so pt_last_error() is "" after a call that succeeded
// (launchRenderKernel returns void like the reference's: the message is its only error channel).
__attribute__((visibility("hidden"))) void pt_clear_error();
