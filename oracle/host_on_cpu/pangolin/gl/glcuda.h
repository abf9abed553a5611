// TEST INFRASTRUCTURE — pangolin/gl/glcuda.h: nothing the reference uses from it is needed on the CPU
#pragma once
#include "gl.h"
