#include "xhe_settings.h"
#include <string.h>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>

extern char **environ;

namespace xhe {
namespace {
typedef std::unordered_map<std::string, std::string> snapshot;
std::atomic<const snapshot *> current{nullptr};
std::mutex load_mu;
}

// A new snapshot replaces the old one, which is kept (readers may still hold pointers into it; reloads are rare and small).
void settings_load() {
    std::lock_guard<std::mutex> lk(load_mu);
    snapshot *s = new snapshot;
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "XHE_", 4)) continue;
        const char *eq = strchr(*e, '=');
        if (eq) (*s)[std::string(*e, eq - *e)] = std::string(eq + 1);
    }
    current.store(s, std::memory_order_release);
}

const char *setting(const char *name) {
    const snapshot *s = current.load(std::memory_order_acquire);
    if (!s) { settings_load(); s = current.load(std::memory_order_acquire); }
    auto it = s->find(name);
    return it == s->end() ? nullptr : it->second.c_str();
}
}  // namespace xhe

extern "C" void xhe_settings_reload(void) { xhe::settings_load(); }
