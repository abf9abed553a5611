// TEST INFRASTRUCTURE — pangolin::GlTexture / GlRenderBuffer / GlFramebuffer as recorders (see ../../gl_record.h)
#pragma once
#include <string>
#include "../../gl_record.h"
namespace pangolin {
struct GlTexture {
  GLuint tid = 0;
  GLint width = 0, height = 0, internal_format = 0;
  GlTexture() {}
  GlTexture(GLint w, GLint h, GLint internal, bool sampling_linear = true, int border = 0, GLenum glformat = GL_RGBA, GLenum gltype = GL_UNSIGNED_BYTE,
            GLvoid* = nullptr)
      : width(w), height(h), internal_format(internal) {
    char b[128];
    snprintf(b, sizeof(b), "texture %dx%d internal=%#x linear=%d fmt=%#x type=%#x", w, h, internal, (int)sampling_linear, glformat, gltype);
    tid = glrec::new_id(b);
    rec("GlTexture -> %u %s", tid, b);
  }
  void Bind() const { glBindTexture(GL_TEXTURE_2D, tid); }
  void Unbind() const { glBindTexture(GL_TEXTURE_2D, 0); }
  void Upload(const void*, GLenum fmt, GLenum type) { rec("GlTexture::Upload %u fmt=%#x type=%#x", tid, fmt, type); }
  void Download(void*, GLenum fmt, GLenum type) const { rec("GlTexture::Download %u fmt=%#x type=%#x", tid, fmt, type); }
  void RenderToViewport(bool flip = false) const { rec("GlTexture::RenderToViewport %u flip=%d", tid, (int)flip); }   // display only (Tools/GUI.h:256)
};
struct GlRenderBuffer {
  GLint width = 0, height = 0;
  GLuint rbid = 0;
  GlRenderBuffer() {}
  GlRenderBuffer(GLint w, GLint h, GLint internal = GL_DEPTH_COMPONENT24) : width(w), height(h) {
    rbid = glrec::new_id("renderbuffer");
    rec("GlRenderBuffer -> %u %dx%d internal=%#x", rbid, w, h, internal);
  }
};
struct GlFramebuffer {
  GLuint fbid = 0;
  unsigned attachments = 0;
  GlFramebuffer() { fbid = glrec::new_id("framebuffer"); }
  GLenum AttachColour(GlTexture& t) { rec("GlFramebuffer %u AttachColour %u <- texture %u", fbid, attachments, t.tid); return attachments++; }
  void AttachDepth(GlRenderBuffer& r) { rec("GlFramebuffer %u AttachDepth renderbuffer %u (%dx%d)", fbid, r.rbid, r.width, r.height); }
  void Bind() const { rec("GlFramebuffer %u Bind (draw buffers 0..%u)", fbid, attachments ? attachments - 1 : 0); }
  void Unbind() const { rec("GlFramebuffer %u Unbind", fbid); }
};
}  // namespace pangolin
