// TEST INFRASTRUCTURE — pangolin::GlSlProgram as a recorder (see ../../gl_record.h)
#pragma once
#include <map>
#include <string>
#include <vector>
#include "gl.h"
namespace pangolin {
enum GlSlShaderType { GlSlVertexShader = 1, GlSlGeometryShader = 2, GlSlFragmentShader = 3 };
class GlSlProgram {
 public:
  GlSlProgram() : prog(glrec::new_id("program")) {}
  bool AddShaderFromFile(GlSlShaderType type, const std::string& filename, const std::map<std::string, std::string>& = {},
                         const std::vector<std::string>& = {}) {
    const std::string base = filename.substr(filename.rfind('/') + 1);
    glrec::S().names[prog] += " " + base;
    rec("program %u AddShaderFromFile type=%d %s", prog, (int)type, base.c_str());
    return true;
  }
  bool Link() { rec("program %u Link:%s", prog, glrec::S().names[prog].substr(7).c_str()); return true; }
  void Bind() { rec("program Bind:%s", glrec::S().names[prog].substr(7).c_str()); }
  void Unbind() { rec("program Unbind"); }
  GLuint ProgramId() const { return prog; }

 protected:
  GLuint prog;
};
}  // namespace pangolin
