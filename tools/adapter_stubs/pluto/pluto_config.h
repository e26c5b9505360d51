// stand-in for the header CMake generates from pluto/src/pluto/pluto_config.h.in (front-end check only)
#pragma once
#define PLUTO_HAVE_HIC 0
#define PLUTO_HAVE_PMR 1
#define PLUTO_HAVE_MDSPAN 0
#define PLUTO_MDSPAN_USE_PAREN_OPERATOR 0
#include "hic/hic_config.h"
#define PLUTO_DEBUGGING 0
