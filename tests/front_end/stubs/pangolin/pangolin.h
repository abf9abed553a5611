// TEST INFRASTRUCTURE — what the reference's GUI front end (Tools/GUI.h, MainController.cpp) needs from <pangolin/pangolin.h> beyond the
// GL wrapper classes the oracle already stands in for (oracle/host_on_cpu/pangolin): windows, views, panels, widgets bound to variables,
// plots.  Declarations with trivial bodies, enough for g++ -fsyntax-only of MainController.cpp where it lies
// (tests/test_front_end_compiles.py); nothing here is ever executed.  Written from the uses in those two files, not from Pangolin.
#pragma once
#include_next <pangolin/pangolin.h>
#include "gl_display.h"
#include <functional>
#include <string>
#include <vector>
namespace pangolin {
struct Params {
  template <typename T> void Set(const std::string&, const T&) {}
};
inline void CreateWindowAndBind(const std::string&, int, int, const Params& = Params()) {}
inline void SetFullscreen(bool) {}
inline bool ShouldQuit() { return false; }
inline void FinishFrame() {}

enum AxisDirection { AxisNone, AxisNegX, AxisX, AxisNegY, AxisY, AxisNegZ, AxisZ };
enum Layout { LayoutOverlay, LayoutVertical, LayoutHorizontal, LayoutEqual, LayoutEqualVertical, LayoutEqualHorizontal };
inline OpenGlMatrix ProjectionMatrix(int, int, double, double, double, double, double, double) { return OpenGlMatrix(); }
inline OpenGlMatrix ModelViewLookAt(double, double, double, double, double, double, AxisDirection) { return OpenGlMatrix(); }
inline OpenGlMatrix operator*(const OpenGlMatrix& a, const OpenGlMatrix&) { return a; }

class OpenGlRenderState {
 public:
  OpenGlRenderState() {}
  OpenGlRenderState(const OpenGlMatrix&, const OpenGlMatrix&) {}
  OpenGlMatrix& GetModelViewMatrix() { return mv; }
  OpenGlMatrix GetModelViewMatrix() const { return mv; }
  OpenGlMatrix GetProjectionModelViewMatrix() const { return mv; }
  OpenGlMatrix GetProjectionMatrix() const { return mv; }
  void SetModelViewMatrix(const OpenGlMatrix& m) { mv = m; }
  void Apply() const {}

 private:
  OpenGlMatrix mv;
};

struct Attach {
  Attach(double = 0) {}
  Attach(int) {}
  static Attach Pix(int) { return Attach(); }
};
struct Viewport { int l = 0, b = 0, w = 0, h = 0; };
struct Handler { virtual ~Handler() {} };
struct Handler3D : Handler {
  explicit Handler3D(OpenGlRenderState&) {}
};
struct View {
  Viewport v;
  View& SetBounds(Attach, Attach, Attach, Attach) { return *this; }
  View& SetBounds(Attach, Attach, Attach, Attach, double) { return *this; }
  View& SetHandler(Handler*) { return *this; }
  View& SetAspect(double) { return *this; }
  View& SetLayout(Layout) { return *this; }
  View& AddDisplay(View&) { return *this; }
  void Activate() const {}
  void Activate(const OpenGlRenderState&) const {}
};
inline View& Display(const std::string&) { static View v; return v; }
inline View& DisplayBase() { static View v; return v; }
inline View& CreatePanel(const std::string&) { static View v; return v; }

template <typename T> struct VarValue {   // what Var<T>::Ref() hands out: gui->pause->Ref().Set(x) (MainController.cpp:108-116, 295)
  T value;
  void Set(const T& v) { value = v; }
  const T& Get() const { return value; }
};
template <typename T> class Var {
 public:
  Var(const std::string&, const T& v = T()) : var{v} {}
  Var(const std::string&, const T& v, bool) : var{v} {}
  Var(const std::string&, const T& v, double, double, bool = false) : var{v} {}
  const T& Get() const { return var.value; }
  operator const T&() const { return var.value; }
  const T* operator->() const { return &var.value; }
  void operator=(const T& v) { var.value = v; }
  VarValue<T>& Ref() { return var; }

 private:
  VarValue<T> var;
};
template <> class Var<std::string> {
 public:
  Var(const std::string&, const char* v = "") : var{v} {}
  Var(const std::string&, const std::string& v) : var{v} {}
  const std::string& Get() const { return var.value; }
  operator const std::string&() const { return var.value; }
  void operator=(const std::string& v) { var.value = v; }
  VarValue<std::string>& Ref() { return var; }

 private:
  VarValue<std::string> var;
};
inline bool Pushed(Var<bool>& button) { const bool was = button.Get(); button = false; return was; }
template <typename T> struct SetVarFunctor {
  SetVarFunctor(const std::string&, T) {}
  void operator()() {}
};
inline void RegisterKeyPressCallback(int, std::function<void(void)>) {}

class DataLog {
 public:
  void SetLabels(const std::vector<std::string>&) {}
  void Log(float) {}
  void Log(float, float) {}
  void Clear() {}
};
class Plotter : public View {
 public:
  Plotter(DataLog*, float = 0, float = 600, float = -1, float = 1, float = 30, float = 0.5f) {}
  void Track(const std::string& = "$i", const std::string& = "") {}
  void ScrollView(float, float) {}
  void ResetView() {}
};
}  // namespace pangolin
