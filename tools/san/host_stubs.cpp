// tools/sanitize.sh only: the device-side entry points that csrc/elm_glue.cpp refers to, so that the host-only sanitizer build of the
// library loads (CPython opens libraries with RTLD_NOW).  None of the CPU tests reaches them; they abort if something does.
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

struct elm_ctx;
struct elm_map;
struct elm_deskew_tables;
struct elm_reg_config;
struct elm_reg_result;

static int die(const char* what) {
    fprintf(stderr, "host-only sanitizer build: %s needs the device library\n", what);
    abort();
    return -2;
}
namespace elm_host {
void* callback_staging(elm_ctx*, unsigned long) { die("callback_staging"); return nullptr; }
int callback_register(elm_ctx*, const elm_map*, const void*, const float*, unsigned long, const elm_deskew_tables*, double, const double*,
                      const elm_reg_config*, elm_reg_result*, unsigned long*, int*) { return die("callback_register"); }
} // namespace elm_host
extern "C" int elm_deskew() { return die("elm_deskew"); }
extern "C" int elm_deskew_prepare() { return die("elm_deskew_prepare"); }
extern "C" int elm_register() { return die("elm_register"); }
extern "C" const char* elm_strerror(int status) { // (the real one lives in elm_api.cpp)
    static char buf[32];
    snprintf(buf, sizeof buf, "status %d", status);
    return buf;
}
extern "C" const char* elm_last_error(const elm_ctx*) { return ""; }
