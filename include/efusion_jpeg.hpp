// JPEG-compressed colour frames of a .klg log (Tools/RawLogReader.cpp:80-109 -> Tools/JPEGLoader.h:40-90).
//
// The reference decodes with the system libjpeg (default settings: ISLOW IDCT, fancy upsampling, RGB output) and then stores
// every pixel with its first and third byte EXCHANGED (JPEGLoader.h:79-84 reads the decoder's row as "bgr" and writes
// rgb[2] = t0, rgb[0] = t2): colour frames that come out of a JPEG arrive R/B-swapped, raw frames do not (quirk Q5 — the
// tracker's intensity and the surfel colours are computed on those bytes as they are).  This decoder reproduces exactly that.
//
// This image ships libjpeg's runtime (libjpeg.so.8) but not its headers, so the library is loaded with dlopen() and the
// handful of ABI facts needed are declared here: the leading fields of jpeg_decompress_struct / jpeg_error_mgr /
// jpeg_source_mgr / jpeg_memory_mgr, which are identical in libjpeg 6b, 7, 8, 9 and libjpeg-turbo.  The two facts that DO
// differ between builds — JPEG_LIB_VERSION and sizeof(jpeg_decompress_struct) — are asked from the library itself:
// jpeg_CreateDecompress() checks both and reports the value it expects through the error manager, so a first call with
// zeros yields them.  Without a libjpeg the reader fails loudly on the first JPEG frame; raw logs are unaffected.
#ifndef EFUSION_JPEG_HPP_
#define EFUSION_JPEG_HPP_
#include <dlfcn.h>
#include <setjmp.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace efusion {

class JpegDecoder {
 public:
  JpegDecoder() {
    for (const char* name : {"libjpeg.so.8", "libjpeg.so.62", "libjpeg.so.9", "libjpeg.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) throw std::runtime_error("JPEG-compressed colour frame, but no libjpeg (libjpeg.so.8 / .62 / .9) can be loaded");
    std_error = (StdError)sym("jpeg_std_error");
    create = (Create)sym("jpeg_CreateDecompress");
    read_header = (ReadHeader)sym("jpeg_read_header");
    calc_dims = (Void1)sym("jpeg_calc_output_dimensions");
    start = (Bool1)sym("jpeg_start_decompress");
    read_scanlines = (ReadScanlines)sym("jpeg_read_scanlines");
    finish = (Bool1)sym("jpeg_finish_decompress");
    destroy = (Void1)sym("jpeg_destroy_decompress");
    resync = (Resync)sym("jpeg_resync_to_restart");
    probe_abi();
  }
  ~JpegDecoder() { if (lib) dlclose(lib); }
  JpegDecoder(const JpegDecoder&) = delete;
  JpegDecoder& operator=(const JpegDecoder&) = delete;

  int libVersion() const { return version; }

  // JPEGLoader::readData: decodes `numBytes` at src into data (width * height * 3 bytes), every pixel stored {t2, t1, t0} of
  // the decoder's {t0, t1, t2}.  max_bytes guards the destination (the reference trusts the stream's own dimensions).
  void readData(const uint8_t* src, size_t numBytes, uint8_t* data, size_t max_bytes) {
    Session S(this);
    S.srcmgr.next_input_byte = src;
    S.srcmgr.bytes_in_buffer = numBytes;
    if (setjmp(S.err.jump)) {
      destroy(S.cinfo());
      throw std::runtime_error("JPEG decoding error (libjpeg message code " + std::to_string(S.err.msg_code()) + ")");
    }
    create(S.cinfo(), version, struct_size);
    S.field<void*>(OFF_SRC) = &S.srcmgr;
    read_header(S.cinfo(), 1);
    calc_dims(S.cinfo());
    start(S.cinfo());
    const unsigned width = S.field<unsigned>(OFF_OUTPUT_WIDTH);
    unsigned height = S.field<unsigned>(OFF_OUTPUT_HEIGHT);
    const int comps = S.field<int>(OFF_OUTPUT_COMPONENTS);
    if (comps != 3) S.fail(this, "colour frame is not a 3-component JPEG");
    if ((size_t)width * height * 3 > max_bytes) S.fail(this, "JPEG frame larger than the configured resolution");
    // (*cinfo.mem->alloc_sarray)((j_common_ptr)&cinfo, JPOOL_IMAGE, width * 4, 1)
    auto mem = S.field<MemoryMgr*>(OFF_MEM);
    uint8_t** buffer = mem->alloc_sarray(S.cinfo(), 1 /* JPOOL_IMAGE */, width * 4, 1);
    for (; height--; data += (size_t)width * 3) {
      read_scanlines(S.cinfo(), buffer, 1);
      const uint8_t* bgr = buffer[0];
      uint8_t* rgb = data;
      for (unsigned i = 0; i < width; i++, bgr += 3, rgb += 3) {
        const uint8_t t0 = bgr[0], t1 = bgr[1], t2 = bgr[2];
        rgb[2] = t0;
        rgb[1] = t1;
        rgb[0] = t2;
      }
    }
    finish(S.cinfo());
    destroy(S.cinfo());
  }

 private:
  // ---- the ABI facts (jpeglib.h of every IJG / libjpeg-turbo release; LP64) ----
  struct ErrorMgr {                       // struct jpeg_error_mgr, followed by our jump buffer
    void (*error_exit)(void* cinfo);      // offset 0
    void (*emit_message)(void*, int);
    void (*output_message)(void*);
    void (*format_message)(void*, char*);
    void (*reset_error_mgr)(void*);
    unsigned char rest[512];              // msg_code (int) at +40, msg_parm.i[8] at +44, tables ...: 168 bytes in all known builds
    jmp_buf jump;
    int msg_code() const { int v; std::memcpy(&v, rest, sizeof(v)); return v; }
    int msg_parm(int k) const { int v; std::memcpy(&v, rest + 4 + 4 * k, sizeof(v)); return v; }
  };
  struct SourceMgr {                      // struct jpeg_source_mgr
    const uint8_t* next_input_byte;
    size_t bytes_in_buffer;
    void (*init_source)(void*);
    int (*fill_input_buffer)(void*);
    void (*skip_input_data)(void*, long);
    int (*resync_to_restart)(void*, int);
    void (*term_source)(void*);
  };
  struct MemoryMgr {                      // struct jpeg_memory_mgr, first three members
    void* (*alloc_small)(void*, int, size_t);
    void* (*alloc_large)(void*, int, size_t);
    uint8_t** (*alloc_sarray)(void*, int, unsigned, unsigned);
  };
  // struct jpeg_decompress_struct: err, mem, progress, client_data (pointers), is_decompressor, global_state (ints), src, then
  // image_width .. enable_2pass_quant (19 ints with one double in between), output_width, output_height, out_color_components,
  // output_components
  enum { OFF_ERR = 0, OFF_MEM = 8, OFF_SRC = 40, OFF_OUTPUT_WIDTH = 136, OFF_OUTPUT_HEIGHT = 140, OFF_OUTPUT_COMPONENTS = 148 };

  using StdError = void* (*)(void*);
  using Create = void (*)(void*, int, size_t);
  using ReadHeader = int (*)(void*, int);
  using Void1 = void (*)(void*);
  using Bool1 = int (*)(void*);
  using ReadScanlines = unsigned (*)(void*, uint8_t**, unsigned);
  using Resync = int (*)(void*, int);

  static void on_error(void* cinfo) {     // error_exit must not return
    ErrorMgr* e;
    std::memcpy(&e, cinfo, sizeof(e));    // cinfo->err is the first member
    longjmp(e->jump, 1);
  }
  static void no_op(void*) {}
  static int fill_input(void* cinfo) {    // the whole frame is in memory: running dry means a truncated stream -> feed an EOI marker
    static const uint8_t eoi[2] = {0xFF, 0xD9};
    SourceMgr* s;
    std::memcpy(&s, (char*)cinfo + OFF_SRC, sizeof(s));
    s->next_input_byte = eoi;
    s->bytes_in_buffer = 2;
    return 1;
  }
  static void skip_input(void* cinfo, long n) {
    SourceMgr* s;
    std::memcpy(&s, (char*)cinfo + OFF_SRC, sizeof(s));
    if (n <= 0) return;
    if ((size_t)n > s->bytes_in_buffer) n = (long)s->bytes_in_buffer;
    s->next_input_byte += n;
    s->bytes_in_buffer -= (size_t)n;
  }

  struct Session {                        // one decode: cinfo (opaque, oversized), error manager, source manager
    std::vector<uint64_t> storage;
    ErrorMgr err;
    SourceMgr srcmgr;
    explicit Session(JpegDecoder* d) : storage(4096 / 8, 0) {
      std::memset(&err, 0, sizeof(err));
      void* e = d->std_error(&err);       // fills the table pointers and the default handlers
      err.error_exit = &JpegDecoder::on_error;
      field<void*>(OFF_ERR) = e;
      srcmgr = SourceMgr{nullptr, 0, &JpegDecoder::no_op, &JpegDecoder::fill_input, &JpegDecoder::skip_input, d->resync, &JpegDecoder::no_op};
    }
    void* cinfo() { return storage.data(); }
    template <typename T>
    T& field(size_t off) { return *reinterpret_cast<T*>(reinterpret_cast<char*>(storage.data()) + off); }
    [[noreturn]] void fail(JpegDecoder* d, const char* what) {
      d->destroy(cinfo());
      throw std::runtime_error(what);
    }
  };

  // jpeg_CreateDecompress(cinfo, version, structsize) checks version, then structsize, and reports what IT has in msg_parm.i[0]
  void probe_abi() {
    for (int round = 0; round < 2; ++round) {
      Session S(this);
      if (setjmp(S.err.jump)) {
        const int expected = S.err.msg_parm(0);
        if (round == 0) version = expected; else struct_size = (size_t)expected;
        continue;
      }
      create(S.cinfo(), round == 0 ? -1 : version, 0);
      throw std::runtime_error("libjpeg accepted an impossible ABI handshake");
    }
    if (version < 60 || version > 100 || struct_size < 400 || struct_size > 4096)
      throw std::runtime_error("libjpeg ABI handshake gave version " + std::to_string(version) + ", struct size " + std::to_string(struct_size));
  }
  void* sym(const char* n) {
    void* p = dlsym(lib, n);
    if (!p) throw std::runtime_error(std::string("libjpeg lacks ") + n);
    return p;
  }

  void* lib = nullptr;
  StdError std_error = nullptr;
  Create create = nullptr;
  ReadHeader read_header = nullptr;
  Void1 calc_dims = nullptr, destroy = nullptr;
  Bool1 start = nullptr, finish = nullptr;
  ReadScanlines read_scanlines = nullptr;
  Resync resync = nullptr;
  int version = 0;
  size_t struct_size = 0;
};

}  // namespace efusion
#endif  // EFUSION_JPEG_HPP_
