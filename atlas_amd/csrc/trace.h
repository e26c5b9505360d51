// roctx ranges with the reference's ATLAS_TRACE labels (TransLocal.cc:948,1107,1159,1418,1446,1529; HaloExchange.h:153,232),
// so that a rocprofv3 --marker-trace timeline of an Atlas run reads the same with this backend.  libroctx64 is bound at
// first use; without it the ranges are no-ops.
#pragma once
#include <dlfcn.h>

namespace atlas_amd {

class TraceRange {
public:
    explicit TraceRange(const char* label) {
        const Api& a = api();
        if (a.push) {
            a.push(label);
            active_ = true;
        }
    }
    ~TraceRange() {
        if (active_) {
            api().pop();
        }
    }
    TraceRange(const TraceRange&)            = delete;
    TraceRange& operator=(const TraceRange&) = delete;

private:
    struct Api {
        int (*push)(const char*) = nullptr;
        int (*pop)()             = nullptr;
    };
    static const Api& api() {
        static const Api a = [] {
            Api x;
            void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            }
            if (h) {
                x.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                x.pop  = (int (*)())dlsym(h, "roctxRangePop");
                if (!x.push || !x.pop) {
                    x.push = nullptr;
                    x.pop  = nullptr;
                }
            }
            return x;
        }();
        return a;
    }
    bool active_ = false;
};

}  // namespace atlas_amd
