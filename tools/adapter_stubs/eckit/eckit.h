// eckit's generated feature header (front-end check only; eckit is not part of /root/reference)
#pragma once
#define eckit_VERSION_STR "1.32.0"
#define eckit_HAVE_MPI 0
