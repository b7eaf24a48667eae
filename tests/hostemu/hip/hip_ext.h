// stand-in for <hip/hip_ext.h>: hipExtLaunchKernelGGL is defined by simt.h (events ignored)
#pragma once
