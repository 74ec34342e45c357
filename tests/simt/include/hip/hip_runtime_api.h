// TEST INFRASTRUCTURE ONLY: see hip_runtime.h in this directory.
#pragma once
#include "hip_runtime.h"
