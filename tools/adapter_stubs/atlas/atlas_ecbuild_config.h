// stand-in for ecbuild's generated configuration header (front-end check only)
#pragma once
#define ATLAS_VERSION_STR "0.44.1"
