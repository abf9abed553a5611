// TEST INFRASTRUCTURE — what Core/GPUTexture.h needs from Pangolin / OpenGL to be parsed: two type names.
#pragma once
#include <string>
typedef unsigned int GLenum;
typedef unsigned int GLuint;
namespace pangolin {
struct GlTexture { GLuint tid = 0; };
}
