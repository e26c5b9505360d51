// stand-in for the header CMake generates from hic/src/hic/hic_config.h.in: the dummy backend (front-end check only)
#pragma once
#define HIC_BACKEND_CUDA 0
#define HIC_BACKEND_HIP 0
#define HIC_BACKEND_DUMMY 1
#define HIC_COMPILER 0
#define HIC_HOST_DEVICE
#define HIC_DEVICE
#define HIC_HOST
#define HIC_GLOBAL
#define HIC_HOST_COMPILE 1
#define HIC_DEVICE_COMPILE 0
