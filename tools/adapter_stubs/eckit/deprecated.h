// eckit's deprecation markers (front-end check only)
#pragma once
#define DEPRECATED(x) [[deprecated(x)]]
#define ECKIT_DEPRECATED(x) [[deprecated(x)]]
