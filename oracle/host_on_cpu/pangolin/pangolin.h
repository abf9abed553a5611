// TEST INFRASTRUCTURE — what the reference's headers need from <pangolin/pangolin.h>: the GL type names and wrapper classes
#pragma once
#include <string>
#include "gl/gl.h"
#include "gl/glsl.h"
#include "display/opengl_render_state.h"
#include "utils/file_utils.h"
