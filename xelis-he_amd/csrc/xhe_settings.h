// The library's XHE_* switches (diagnostics, A/B measurements, tests; none is needed in production).  The environment is read
// once, when the first context is created: a switch set afterwards has no effect until xhe_settings_reload() (include/xhe.h),
// which exists for tests and measurement tools that change switches inside one process.
#pragma once
namespace xhe {
// the value the variable had at that moment, or nullptr if it was not set; the pointer stays valid for the life of the process
const char *setting(const char *name);
void settings_load();
}
