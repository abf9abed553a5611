// TEST INFRASTRUCTURE — what the reference's front end (Main.cpp, MainController.cpp, Tools/RawLogReader.cpp, Core/Utils/Parse.cpp, compiled where
// they lie by `make reffrontend`) needs besides libefusion.so to become an executable without a window, a sensor or CUDA:
//   * GPUTexture (the GUI's FXAA colour target): Core/GPUTexture.cpp:28-46 without the CUDA registration, over the recorder's GlTexture;
//   * LiveLogReader (OpenNI2 / RealSense capture): no sensor in this build — constructing one says so and exits;
//   * GroundTruthOdometry (-p <poses>): not in this build.
// The resulting binary, oracle/_ref/reference_front_end, is the reference's own run loop (MainController::run, minus the 13 OpenGL
// statements INTEGRATION.md lists, taken out in a pipe) driving THIS repository's library: `reference_front_end -l log.klg -q` replays a
// log and writes log.klg.freiburg from ~ElasticFusion, like the reference's ElasticFusion binary with -q.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "MainController.h"

const std::string GPUTexture::RGB = "RGB";
const std::string GPUTexture::DEPTH_RAW = "DEPTH";
const std::string GPUTexture::DEPTH_FILTERED = "DEPTH_FILTERED";
const std::string GPUTexture::DEPTH_METRIC = "DEPTH_METRIC";
const std::string GPUTexture::DEPTH_METRIC_FILTERED = "DEPTH_METRIC_FILTERED";
const std::string GPUTexture::DEPTH_NORM = "DEPTH_NORM";
GPUTexture::GPUTexture(const int w, const int h, const GLenum internalFormat_, const GLenum format_, const GLenum dataType_, const bool draw_)
    : texture(new pangolin::GlTexture(w, h, internalFormat_, draw_, 0, format_, dataType_)), cudaRes(nullptr), draw(draw_), width(w), height(h),
      internalFormat(internalFormat_), format(format_), dataType(dataType_) {}
GPUTexture::~GPUTexture() { delete texture; }

namespace {
[[noreturn]] void not_in_this_build(const char* what) {
  std::fprintf(stderr, "reference_front_end: %s is not part of this build (give a log with -l <file.klg>)\n", what);
  std::exit(3);
}
}  // namespace

LiveLogReader::LiveLogReader(std::string file, bool flipColors, CameraType) : LogReader(file, flipColors), cam(nullptr), lastFrameTime(-1), lastGot(-1) {
  not_in_this_build("live capture (OpenNI2 / RealSense)");
}
LiveLogReader::~LiveLogReader() {}
void LiveLogReader::getNext() {}
int LiveLogReader::getNumFrames() { return 0; }
bool LiveLogReader::hasMore() { return false; }
const std::string LiveLogReader::getFile() { return std::string(); }
void LiveLogReader::setAuto(bool) {}

GroundTruthOdometry::GroundTruthOdometry(const std::string&) : last_utime(0) { not_in_this_build("ground-truth odometry (-p)"); }
GroundTruthOdometry::~GroundTruthOdometry() {}
Eigen::Matrix4f GroundTruthOdometry::getTransformation(uint64_t) { return Eigen::Matrix4f::Identity(); }
