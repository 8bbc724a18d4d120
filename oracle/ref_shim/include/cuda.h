// stand-in for the CUDA toolkit header of the same name: see dgemu.h (TEST INFRASTRUCTURE ONLY)
#pragma once
#include "dgemu.h"
