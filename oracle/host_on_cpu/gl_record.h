// TEST INFRASTRUCTURE — "OpenGL as a tape recorder": the OpenGL entry points and the Pangolin wrapper classes that the reference's
// map-side HOST code uses (Core/IndexMap.cpp, GlobalModel.cpp, Shaders/{FillIn,ComputePack,FeedbackBuffer,Resize}.cpp, Shaders.h),
// implemented so that those files can be compiled where they lie and, when run, leave a transcript of what they asked the GL to
// do: programs (by shader file), uniforms by NAME with their values, texture-unit bindings, attachments, viewport, clears,
// capabilities, vertex attribute layouts, transform-feedback set-up and the draw calls, in order.  Nothing is rendered: the
// shaders themselves are compiled and executed by oracle/glsl_on_cpu.  The transcript is what tests compare with the pass
// parameters the oracle and oracle/ref_glsl_bridge.cpp use (tests/test_oracle_vs_reference_glhost.py).
// Written from scratch; enum values are arbitrary but distinct, efgl_constant() resolves them by name.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef unsigned int GLenum;
typedef unsigned int GLuint;
typedef int GLint;
typedef int GLsizei;
typedef unsigned int GLbitfield;
typedef float GLfloat;
typedef unsigned char GLboolean;
typedef void GLvoid;
typedef ptrdiff_t GLsizeiptr;
typedef ptrdiff_t GLintptr;

namespace glrec {
struct State {
  std::string log;
  std::map<std::string, GLenum> constants;
  std::map<GLuint, std::string> names;                          // object id -> label (program files, "tex WxH fmt", ...)
  std::map<GLint, std::pair<GLuint, std::string>> uniform_loc;  // location -> (program, name)
  GLuint next_id = 1;
  GLuint query_result = 0;                                      // what glGetQueryObjectuiv hands back (set by the bridge)
  int query_once = -1;                                          // >= 0: handed back by the NEXT query only, then query_result again
  int readpixels_fill = -1;                                     // >= 0: glReadPixels fills its destination with this byte (-1: with zeros)
  std::vector<std::vector<unsigned char>> readpixels_queue;     // if not empty: successive glReadPixels calls copy these out, in order
  std::vector<int> query_queue;                                 // if not empty: successive queries hand these back, in order
  std::vector<unsigned char> buffer_data;                       // what glGetBufferSubData copies out (e.g. the surfel map)
};
inline State& S() { static State s; return s; }
inline void rec(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  S().log += buf;
  S().log += '\n';
}
inline GLuint new_id(const std::string& label) { const GLuint id = S().next_id++; S().names[id] = label; return id; }
struct Reg { Reg(const char* n, GLenum v) { S().constants[n] = v; } };
}  // namespace glrec

#define GLREC_CONST(name, val) static const GLenum name = (val); static const glrec::Reg glrec_reg_##name(#name, (val));
GLREC_CONST(GL_FALSE, 0) GLREC_CONST(GL_TRUE, 1)
GLREC_CONST(GL_POINTS, 0x0000)
GLREC_CONST(GL_DEPTH_BUFFER_BIT, 0x0100) GLREC_CONST(GL_COLOR_BUFFER_BIT, 0x4000) GLREC_CONST(GL_VIEWPORT_BIT, 0x0800)
GLREC_CONST(GL_TEXTURE_2D, 0x0DE1) GLREC_CONST(GL_UNSIGNED_BYTE, 0x1401) GLREC_CONST(GL_UNSIGNED_SHORT, 0x1403)
GLREC_CONST(GL_UNSIGNED_INT, 0x1405) GLREC_CONST(GL_FLOAT, 0x1406) GLREC_CONST(GL_RGB, 0x1907) GLREC_CONST(GL_RGBA, 0x1908)
GLREC_CONST(GL_LUMINANCE, 0x1909) GLREC_CONST(GL_RGBA32F, 0x8814) GLREC_CONST(GL_LUMINANCE32F_ARB, 0x8818)
GLREC_CONST(GL_LUMINANCE16UI_EXT, 0x8D7A) GLREC_CONST(GL_LUMINANCE32UI_EXT, 0x8D74) GLREC_CONST(GL_LUMINANCE_INTEGER_EXT, 0x8D9C)
GLREC_CONST(GL_DEPTH_COMPONENT24, 0x81A6)
GLREC_CONST(GL_TEXTURE0, 0x84C0) GLREC_CONST(GL_TEXTURE1, 0x84C1) GLREC_CONST(GL_TEXTURE2, 0x84C2) GLREC_CONST(GL_TEXTURE3, 0x84C3)
GLREC_CONST(GL_TEXTURE4, 0x84C4) GLREC_CONST(GL_TEXTURE5, 0x84C5) GLREC_CONST(GL_TEXTURE6, 0x84C6)
GLREC_CONST(GL_ARRAY_BUFFER, 0x8892) GLREC_CONST(GL_STREAM_DRAW, 0x88E0) GLREC_CONST(GL_STATIC_DRAW, 0x88E4)
GLREC_CONST(GL_COPY_READ_BUFFER, 0x8F36) GLREC_CONST(GL_COPY_WRITE_BUFFER, 0x8F37)
GLREC_CONST(GL_POINT_SPRITE, 0x8861) GLREC_CONST(GL_PROGRAM_POINT_SIZE, 0x8642) GLREC_CONST(GL_RASTERIZER_DISCARD, 0x8C89)
GLREC_CONST(GL_INTERLEAVED_ATTRIBS, 0x8C8C) GLREC_CONST(GL_TRANSFORM_FEEDBACK_BUFFER, 0x8C8E)
GLREC_CONST(GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN, 0x8C88) GLREC_CONST(GL_TRANSFORM_FEEDBACK, 0x8E22)
GLREC_CONST(GL_QUERY_RESULT, 0x8866) GLREC_CONST(GL_DEPTH_TEST, 0x0B71)

using glrec::rec;
inline void glFinish() {}
inline void glBindBuffer(GLenum target, GLuint b) { rec("glBindBuffer %#x %u", target, b); }
inline void glBindTexture(GLenum target, GLuint t) { rec("glBindTexture %#x %u", target, t); }
inline void glActiveTexture(GLenum unit) { rec("glActiveTexture %u", unit - GL_TEXTURE0); }
inline void glVertexAttribPointer(GLuint i, GLint size, GLenum type, GLboolean norm, GLsizei stride, const void* off) {
  rec("glVertexAttribPointer %u size=%d type=%#x norm=%d stride=%d offset=%ld", i, size, type, (int)norm, stride, (long)(intptr_t)off);
}
inline void glEnableVertexAttribArray(GLuint i) { rec("glEnableVertexAttribArray %u", i); }
inline void glDisableVertexAttribArray(GLuint i) { rec("glDisableVertexAttribArray %u", i); }
inline void glViewport(GLint x, GLint y, GLsizei w, GLsizei h) { rec("glViewport %d %d %d %d", x, y, w, h); }
inline void glPushAttrib(GLbitfield m) { rec("glPushAttrib %#x", m); }
inline void glPopAttrib() { rec("glPopAttrib"); }
inline void glClearColor(GLfloat r, GLfloat g, GLfloat b, GLfloat a) { rec("glClearColor %.9g %.9g %.9g %.9g", r, g, b, a); }
inline void glClear(GLbitfield m) { rec("glClear %#x", m); }
inline void glEnable(GLenum c) { rec("glEnable %#x", c); }
inline void glDisable(GLenum c) { rec("glDisable %#x", c); }
inline void glDrawArrays(GLenum mode, GLint first, GLsizei count) { rec("glDrawArrays %#x %d %d", mode, first, count); }
inline void glDrawTransformFeedback(GLenum mode, GLuint tf) { rec("glDrawTransformFeedback %#x %u", mode, tf); }
inline void glGenBuffers(GLsizei n, GLuint* b) { for (int i = 0; i < n; ++i) { b[i] = glrec::new_id("buffer"); rec("glGenBuffers -> %u", b[i]); } }
inline void glDeleteBuffers(GLsizei, const GLuint*) {}
inline void glBufferData(GLenum target, GLsizeiptr size, const void* data, GLenum usage) {
  rec("glBufferData %#x size=%ld data=%s usage=%#x", target, (long)size, data ? "host" : "null", usage);
}
inline void glBindBufferBase(GLenum target, GLuint index, GLuint b) { rec("glBindBufferBase %#x %u %u", target, index, b); }
inline void glGenTransformFeedbacks(GLsizei n, GLuint* ids) { for (int i = 0; i < n; ++i) { ids[i] = glrec::new_id("tf"); rec("glGenTransformFeedbacks -> %u", ids[i]); } }
inline void glDeleteTransformFeedbacks(GLsizei, const GLuint*) {}
inline void glBindTransformFeedback(GLenum target, GLuint id) { rec("glBindTransformFeedback %#x %u", target, id); }
inline void glBeginTransformFeedback(GLenum mode) { rec("glBeginTransformFeedback %#x", mode); }
inline void glEndTransformFeedback() { rec("glEndTransformFeedback"); }
inline GLint glGetVaryingLocationNV(GLuint prog, const char* name) {
  const GLint loc = (GLint)glrec::new_id(std::string("varying ") + name);
  rec("glGetVaryingLocationNV prog=%u %s -> %d", prog, name, loc);
  return loc;
}
inline void glTransformFeedbackVaryingsNV(GLuint prog, GLsizei n, const GLint* locs, GLenum mode) {
  std::string s;
  for (int i = 0; i < n; ++i) s += " " + glrec::S().names[(GLuint)locs[i]].substr(8);
  rec("glTransformFeedbackVaryingsNV prog=%u mode=%#x:%s", prog, mode, s.c_str());
}
inline void glGenQueries(GLsizei n, GLuint* ids) { for (int i = 0; i < n; ++i) ids[i] = glrec::new_id("query"); }
inline void glDeleteQueries(GLsizei, const GLuint*) {}
inline void glBeginQuery(GLenum target, GLuint id) { rec("glBeginQuery %#x %u", target, id); }
inline void glEndQuery(GLenum target) { rec("glEndQuery %#x", target); }
inline void glGetQueryObjectuiv(GLuint id, GLenum pname, GLuint* out) {
  if (!glrec::S().query_queue.empty()) {
    *out = (GLuint)glrec::S().query_queue.front();
    glrec::S().query_queue.erase(glrec::S().query_queue.begin());
  } else {
    *out = glrec::S().query_once >= 0 ? (GLuint)glrec::S().query_once : glrec::S().query_result;
  }
  glrec::S().query_once = -1; rec("glGetQueryObjectuiv %u %#x -> %u", id, pname, *out);
}
inline void glReadPixels(GLint x, GLint y, GLsizei w, GLsizei h, GLenum fmt, GLenum type, void* dst) {
  rec("glReadPixels %d %d %d %d fmt=%#x type=%#x", x, y, w, h, fmt, type);
  if (!glrec::S().readpixels_queue.empty() && dst) {
    const std::vector<unsigned char>& b = glrec::S().readpixels_queue.front();
    memcpy(dst, b.data(), b.size());
    glrec::S().readpixels_queue.erase(glrec::S().readpixels_queue.begin());
  } else if (dst) {   // nothing scripted: a cleared framebuffer (never the destination's uninitialised heap contents)
    const int ch = fmt == GL_RGB ? 3 : fmt == GL_RGBA ? 4 : 1;
    const int bytes = type == GL_FLOAT || type == GL_UNSIGNED_INT ? 4 : type == GL_UNSIGNED_SHORT ? 2 : 1;
    memset(dst, glrec::S().readpixels_fill >= 0 ? glrec::S().readpixels_fill : 0, (size_t)w * h * ch * bytes);
  }
}
inline void glGetBufferSubData(GLenum target, GLintptr off, GLsizeiptr size, void* dst) {
  rec("glGetBufferSubData %#x %ld %ld", target, (long)off, (long)size);
  const std::vector<unsigned char>& b = glrec::S().buffer_data;
  if (dst && !b.empty() && (size_t)off < b.size()) memcpy(dst, b.data() + off, std::min((size_t)size, b.size() - (size_t)off));
}
inline void glCopyBufferSubData(GLenum r, GLenum w, GLintptr ro, GLintptr wo, GLsizeiptr size) { rec("glCopyBufferSubData %#x %#x %ld %ld %ld", r, w, (long)ro, (long)wo, (long)size); }
inline void glTexSubImage2D(GLenum target, GLint level, GLint x, GLint y, GLsizei w, GLsizei h, GLenum fmt, GLenum type, const void*) {
  rec("glTexSubImage2D %#x %d %d %d %d %d fmt=%#x type=%#x", target, level, x, y, w, h, fmt, type);
}
inline GLint glGetUniformLocation(GLuint prog, const char* name) {
  const GLint loc = (GLint)glrec::S().next_id++;
  glrec::S().uniform_loc[loc] = {prog, name};
  return loc;
}
inline const char* glrec_uname(GLint loc) { return glrec::S().uniform_loc[loc].second.c_str(); }
inline void glUniform1i(GLint loc, GLint v) { rec("uniform %s int %d", glrec_uname(loc), v); }
inline void glUniform1f(GLint loc, GLfloat v) { rec("uniform %s float %.9g", glrec_uname(loc), v); }
inline void glUniform2f(GLint loc, GLfloat a, GLfloat b) { rec("uniform %s vec2 %.9g %.9g", glrec_uname(loc), a, b); }
inline void glUniform3f(GLint loc, GLfloat a, GLfloat b, GLfloat c) { rec("uniform %s vec3 %.9g %.9g %.9g", glrec_uname(loc), a, b, c); }
inline void glUniform4f(GLint loc, GLfloat a, GLfloat b, GLfloat c, GLfloat d) { rec("uniform %s vec4 %.9g %.9g %.9g %.9g", glrec_uname(loc), a, b, c, d); }
inline void glUniformMatrix4fv(GLint loc, GLsizei, GLboolean transpose, const GLfloat* m) {
  std::string s;
  char b[32];
  for (int i = 0; i < 16; ++i) { snprintf(b, sizeof(b), " %.9g", m[i]); s += b; }
  rec("uniform %s mat4 transpose=%d column-major:%s", glrec_uname(loc), (int)transpose, s.c_str());
}
