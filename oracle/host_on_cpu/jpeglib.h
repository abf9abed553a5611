/* TEST INFRASTRUCTURE — the part of libjpeg's API that Tools/JPEGLoader.h uses, declared so that the REFERENCE's own reader
 * (Tools/RawLogReader.cpp + JPEGLoader.h, compiled where they lie: oracle/Makefile `refklg`) decodes JPEG colour frames with the
 * system's libjpeg RUNTIME (libjpeg.so.8 is in this image, its headers are not).  Only the leading members of the structures are
 * spelled out (identical in IJG 6b..9 and libjpeg-turbo); the tails are padding large enough for any build, and the two numbers
 * jpeg_create_decompress() must pass — JPEG_LIB_VERSION and sizeof(struct jpeg_decompress_struct) as the LIBRARY was built — are
 * asked from the library (it reports what it expects through the error manager when handed zeros).  Written for this repository. */
#ifndef EFR_JPEGLIB_STUB_H_
#define EFR_JPEGLIB_STUB_H_
#include <dlfcn.h>
#include <setjmp.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#define TRUE 1
#define FALSE 0
#define JPOOL_IMAGE 1
typedef int boolean;
typedef unsigned char JSAMPLE;
typedef JSAMPLE* JSAMPROW;
typedef JSAMPROW* JSAMPARRAY;
typedef unsigned int JDIMENSION;
struct jpeg_common_struct;
struct jpeg_decompress_struct;
typedef struct jpeg_common_struct* j_common_ptr;
typedef struct jpeg_decompress_struct* j_decompress_ptr;
struct jpeg_error_mgr {
  void (*error_exit)(j_common_ptr);
  void (*emit_message)(j_common_ptr, int);
  void (*output_message)(j_common_ptr);
  void (*format_message)(j_common_ptr, char*);
  void (*reset_error_mgr)(j_common_ptr);
  int msg_code;
  union { int i[8]; char s[80]; } msg_parm;
  char tail_[512];
};
struct jpeg_memory_mgr {
  void* (*alloc_small)(j_common_ptr, int, size_t);
  void* (*alloc_large)(j_common_ptr, int, size_t);
  JSAMPARRAY (*alloc_sarray)(j_common_ptr, int, JDIMENSION, JDIMENSION);
};
struct jpeg_source_mgr {
  const unsigned char* next_input_byte;
  size_t bytes_in_buffer;
  void (*init_source)(j_decompress_ptr);
  boolean (*fill_input_buffer)(j_decompress_ptr);
  void (*skip_input_data)(j_decompress_ptr, long);
  boolean (*resync_to_restart)(j_decompress_ptr, int);
  void (*term_source)(j_decompress_ptr);
};
struct jpeg_common_struct { struct jpeg_error_mgr* err; struct jpeg_memory_mgr* mem; void* progress; void* client_data; boolean is_decompressor; int global_state; };
struct jpeg_decompress_struct {
  struct jpeg_error_mgr* err;
  struct jpeg_memory_mgr* mem;
  void* progress;
  void* client_data;
  boolean is_decompressor;
  int global_state;
  struct jpeg_source_mgr* src;
  JDIMENSION image_width, image_height;
  int num_components, jpeg_color_space, out_color_space;
  unsigned int scale_num, scale_denom;
  double output_gamma;
  boolean buffered_image, raw_data_out;
  int dct_method;
  boolean do_fancy_upsampling, do_block_smoothing, quantize_colors;
  int dither_mode;
  boolean two_pass_quantize;
  int desired_number_of_colors;
  boolean enable_1pass_quant, enable_external_quant, enable_2pass_quant;
  JDIMENSION output_width, output_height;
  int out_color_components, output_components;
  char tail_[4096];
};
static inline void* efr_jpeg_sym(const char* name) {
  static void* lib = NULL;
  if (!lib) {
    const char* names[] = {"libjpeg.so.8", "libjpeg.so.62", "libjpeg.so.9", "libjpeg.so"};
    for (int i = 0; i < 4 && !lib; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "oracle/host_on_cpu/jpeglib.h: no libjpeg runtime to load\n"); abort(); }
  }
  void* p = dlsym(lib, name);
  if (!p) { fprintf(stderr, "oracle/host_on_cpu/jpeglib.h: libjpeg lacks %s\n", name); abort(); }
  return p;
}
static inline struct jpeg_error_mgr* jpeg_std_error(struct jpeg_error_mgr* e) {
  return ((struct jpeg_error_mgr* (*)(struct jpeg_error_mgr*))efr_jpeg_sym("jpeg_std_error"))(e);
}
/* the handshake: what the loaded library calls JPEG_LIB_VERSION and sizeof(struct jpeg_decompress_struct) */
static jmp_buf efr_jpeg_jump_;
static inline void efr_jpeg_probe_exit_(j_common_ptr) { longjmp(efr_jpeg_jump_, 1); }
static inline void efr_jpeg_abi(int* version, size_t* size) {
  static int v = 0;
  static size_t s = 0;
  if (!v) {
    void (*create)(j_decompress_ptr, int, size_t) = (void (*)(j_decompress_ptr, int, size_t))efr_jpeg_sym("jpeg_CreateDecompress");
    for (int round = 0; round < 2; ++round) {
      static struct jpeg_decompress_struct probe;
      static struct jpeg_error_mgr em;
      probe.err = jpeg_std_error(&em);
      em.error_exit = efr_jpeg_probe_exit_;
      if (setjmp(efr_jpeg_jump_)) {
        if (round == 0) v = em.msg_parm.i[0]; else s = (size_t)em.msg_parm.i[0];
        continue;
      }
      create(&probe, round == 0 ? -1 : v, 0);
    }
  }
  *version = v;
  *size = s;
}
static inline void jpeg_create_decompress(j_decompress_ptr c) {
  int v; size_t s;
  efr_jpeg_abi(&v, &s);
  ((void (*)(j_decompress_ptr, int, size_t))efr_jpeg_sym("jpeg_CreateDecompress"))(c, v, s);
}
static inline boolean jpeg_resync_to_restart(j_decompress_ptr c, int d) { return ((boolean (*)(j_decompress_ptr, int))efr_jpeg_sym("jpeg_resync_to_restart"))(c, d); }
static inline int jpeg_read_header(j_decompress_ptr c, boolean r) { return ((int (*)(j_decompress_ptr, boolean))efr_jpeg_sym("jpeg_read_header"))(c, r); }
static inline void jpeg_calc_output_dimensions(j_decompress_ptr c) { ((void (*)(j_decompress_ptr))efr_jpeg_sym("jpeg_calc_output_dimensions"))(c); }
static inline boolean jpeg_start_decompress(j_decompress_ptr c) { return ((boolean (*)(j_decompress_ptr))efr_jpeg_sym("jpeg_start_decompress"))(c); }
static inline JDIMENSION jpeg_read_scanlines(j_decompress_ptr c, JSAMPARRAY b, JDIMENSION n) {
  return ((JDIMENSION (*)(j_decompress_ptr, JSAMPARRAY, JDIMENSION))efr_jpeg_sym("jpeg_read_scanlines"))(c, b, n);
}
static inline boolean jpeg_finish_decompress(j_decompress_ptr c) { return ((boolean (*)(j_decompress_ptr))efr_jpeg_sym("jpeg_finish_decompress"))(c); }
static inline void jpeg_destroy_decompress(j_decompress_ptr c) { ((void (*)(j_decompress_ptr))efr_jpeg_sym("jpeg_destroy_decompress"))(c); }
#endif
