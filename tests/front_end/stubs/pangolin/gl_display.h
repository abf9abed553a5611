// TEST INFRASTRUCTURE — the OpenGL entry points and constants that only the reference's DISPLAY code calls (Tools/GUI.h, MainController.cpp)
// and the oracle's GL tape recorder (oracle/host_on_cpu/gl_record.h: the map-side passes) therefore does not have.  Declarations with empty
// bodies for g++ -fsyntax-only; enum values are arbitrary.
#pragma once
#include <pangolin/gl/gl.h>
static const GLenum GL_UNPACK_ALIGNMENT = 0xD001, GL_PACK_ALIGNMENT = 0xD002, GL_LESS = 0xD003, GL_READ_FRAMEBUFFER = 0xD004, GL_DRAW_FRAMEBUFFER = 0xD005,
                    GL_NEAREST = 0xD006;
inline void glPixelStorei(GLenum, GLint) {}
inline void glDepthMask(GLboolean) {}
inline void glDepthFunc(GLenum) {}
inline void glGetIntegerv(GLenum, GLint*) {}
inline void glBindFramebuffer(GLenum, GLuint) {}
inline void glBlitFramebuffer(GLint, GLint, GLint, GLint, GLint, GLint, GLint, GLint, GLbitfield, GLenum) {}
inline void glColor3f(GLfloat, GLfloat, GLfloat) {}
