// Every environment switch the library reads goes through env_get() and is listed in ONE table (env.cpp), which is also what
// atlas_amd__effective_config() prints and INTEGRATION.md's table is generated from (tools/gen_env_table.py;
// tests/test_env_switches.py keeps sources, table and document in step).  A library loaded into Atlas inherits the caller's
// environment:
//   * atlas_amd__set_ignore_env(1) -- what the adapter plugin calls at start-up -- or ATLAS_AMD_IGNORE_ENV=1 makes every switch
//     read as unset: the library runs its default configuration whatever the environment holds;
//   * development switches (class "dev": access ablations, single-class runs, probes, the kernels that live in
//     tools/experiments) exist only in builds with -DATLAS_AMD_DEV_SWITCHES (make dev / make experiments): in the product library
//     env_get() answers "unset" for them without looking at the environment, and says so once on stderr if one is set.
#pragma once

namespace atlas_amd {

enum class EnvClass { tuning, behaviour, test_hook, dev };

struct EnvSwitch {
    const char* name;
    EnvClass cls;
    const char* default_value;   // what the library does when the switch is unset
    const char* what;
};

const EnvSwitch* env_switches(int* count);
// value of a switch of the table (nullptr: unset / ignored / compiled out).  A name missing from the table is a programming error:
// it aborts in builds with assertions and is treated as unset otherwise.
const char* env_get(const char* name);
void env_set_ignore(bool on);   // process-wide
bool env_ignored();             // atlas_amd__set_ignore_env(1) or ATLAS_AMD_IGNORE_ENV=<non-zero>
const char* env_class_name(EnvClass c);
bool env_dev_switches_compiled_in();

}  // namespace atlas_amd
