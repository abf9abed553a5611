// .klg log reader of the headless front-end — the file format and the frame-iteration protocol of the reference's
// Tools/RawLogReader.cpp:22-109 (int32 numFrames; per frame int64 timestamp, int32 depthSize, int32 imageSize, depth bytes raw u16
// or zlib, colour bytes raw RGB8 or one JPEG image — decoded through the system libjpeg, loaded at the first such frame, with the
// R/B exchange of the reference's JPEGLoader.h:79-84, see efusion_jpeg.hpp).  Header-only, written from scratch; libefusion.so also
// exports it as a C API (efk_*, bottom of this file) for non-C++ hosts and tests.
//
// Protocol, as the reference's: `while (r.hasMore()) { r.getNext(); use r.rgb / r.depth / r.timestamp; }`.
// hasMore() is `currentFrame + 1 < numFrames` (RawLogReader.cpp:127-129): the reference's run loop therefore never delivers the
// LAST frame of a log.  That is kept by default — a trajectory written by this front-end lines up with one written by the
// reference — and can be switched off (deliverLastFrame) to consume every frame.
#ifndef EFUSION_KLG_HPP_
#define EFUSION_KLG_HPP_
#include <zlib.h>

#include <cstdint>
#include <memory>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "efusion_jpeg.hpp"

namespace efusion {

class KlgReader {
 public:
  std::vector<uint8_t> depth, rgb;   // the current frame: W*H u16 millimetres, W*H*3 RGB8
  int64_t timestamp = 0;
  int currentFrame = 0;
  bool deliverLastFrame = false;
  bool flipColors = false;           // RawLogReader's flipColors: swap R and B after reading (RawLogReader.cpp:101-105)

  KlgReader(const std::string& file, int width, int height) : numPixels(width * height) {
    fp = std::fopen(file.c_str(), "rb");
    if (!fp) throw std::runtime_error("cannot open " + file);
    if (std::fread(&numFrames, sizeof(int32_t), 1, fp) != 1) throw std::runtime_error("empty log");
    depthRead.resize((size_t)numPixels * 2 + 65536);
    imageRead.resize((size_t)numPixels * 3 + 65536);
    depth.resize((size_t)numPixels * 2);
    rgb.resize((size_t)numPixels * 3);
  }
  ~KlgReader() { if (fp) std::fclose(fp); }
  KlgReader(const KlgReader&) = delete;
  KlgReader& operator=(const KlgReader&) = delete;

  int getNumFrames() const { return numFrames; }
  bool hasMore() const { return deliverLastFrame ? currentFrame < numFrames : currentFrame + 1 < numFrames; }
  void getNext() {
    int32_t depthSize = 0, imageSize = 0;
    if (std::fread(&timestamp, sizeof(int64_t), 1, fp) != 1 || std::fread(&depthSize, sizeof(int32_t), 1, fp) != 1 ||
        std::fread(&imageSize, sizeof(int32_t), 1, fp) != 1)
      throw std::runtime_error("truncated log header");
    if (depthSize < 0 || (size_t)depthSize > depthRead.size() || imageSize < 0 || (size_t)imageSize > imageRead.size())
      throw std::runtime_error("frame larger than the configured resolution");
    if (depthSize && std::fread(depthRead.data(), depthSize, 1, fp) != 1) throw std::runtime_error("truncated depth");
    if (imageSize && std::fread(imageRead.data(), imageSize, 1, fp) != 1) throw std::runtime_error("truncated image");
    if (depthSize == numPixels * 2) {
      std::memcpy(depth.data(), depthRead.data(), depth.size());
    } else {
      unsigned long len = depth.size();
      if (uncompress(depth.data(), &len, depthRead.data(), depthSize) != Z_OK) throw std::runtime_error("zlib depth frame corrupt");
    }
    if (imageSize == numPixels * 3) std::memcpy(rgb.data(), imageRead.data(), rgb.size());
    else if (imageSize == 0) std::memset(rgb.data(), 0, rgb.size());
    else {   // RawLogReader.cpp:94-96: anything else is a JPEG image
      if (!jpeg) jpeg.reset(new JpegDecoder());
      jpeg->readData(imageRead.data(), (size_t)imageSize, rgb.data(), rgb.size());
    }
    if (flipColors)
      for (size_t i = 0; i + 2 < rgb.size(); i += 3) std::swap(rgb[i], rgb[i + 2]);
    ++currentFrame;
  }

 private:
  FILE* fp = nullptr;
  int32_t numFrames = 0;
  int numPixels;
  std::vector<uint8_t> depthRead, imageRead;
  std::unique_ptr<JpegDecoder> jpeg;   // created at the first JPEG-compressed frame
};

}  // namespace efusion

// C API (libefusion.so): 0 / NULL on failure, message through efk_last_error()
extern "C" {
void* efk_open(const char* file, int width, int height, int deliver_last_frame, int flip_colors);
void efk_close(void* reader);
int efk_num_frames(void* reader);
int efk_has_more(void* reader);
int efk_next(void* reader, int64_t* timestamp, uint16_t* depth, uint8_t* rgb);   // 1 on success
const char* efk_last_error(void);
}
#endif  // EFUSION_KLG_HPP_
