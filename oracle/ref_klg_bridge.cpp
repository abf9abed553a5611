// TEST INFRASTRUCTURE — drives the REFERENCE's own .klg reader, Tools/RawLogReader.cpp (+ LogReader.h, JPEGLoader.h as headers),
// compiled from /root/reference where it lies against host_on_cpu/ (oracle/Makefile, target `refklg` -> _ref/libefr_klg.so), so that
// the product's reader (include/efusion_klg.hpp) can be compared with it frame by frame.  This file is ours.
#include <cstdint>
#include <cstring>

#include "RawLogReader.h"

extern "C" {
void* efrk_open(const char* file, int width, int height, int flipColors) {
  Resolution::getInstance(width, height);   // process-wide singleton of the reference: first call fixes the size
  if (Resolution::getInstance().width() != width || Resolution::getInstance().height() != height) return nullptr;
  return new RawLogReader(file, flipColors != 0);
}
void efrk_close(void* r) { delete (RawLogReader*)r; }
int efrk_num_frames(void* r) { return ((RawLogReader*)r)->getNumFrames(); }
int efrk_has_more(void* r) { return ((RawLogReader*)r)->hasMore() ? 1 : 0; }
int efrk_next(void* p, int64_t* timestamp, uint16_t* depth, uint8_t* rgb) {
  RawLogReader* r = (RawLogReader*)p;
  r->getNext();
  const int n = Resolution::getInstance().numPixels();
  *timestamp = r->timestamp;
  std::memcpy(depth, r->depth, (size_t)n * 2);
  std::memcpy(rgb, r->rgb, (size_t)n * 3);
  return 1;
}
}
