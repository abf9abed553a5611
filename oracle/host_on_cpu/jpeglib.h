/* TEST INFRASTRUCTURE — enough of the libjpeg declarations for Tools/JPEGLoader.h to be PARSED (libjpeg's headers are absent from
 * this image).  Nothing here decodes: the functions abort.  The reference's reader only reaches them for JPEG-compressed colour
 * frames, which the logs used in the tests do not contain. */
#ifndef EFR_JPEGLIB_STUB_H_
#define EFR_JPEGLIB_STUB_H_
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
struct jpeg_error_mgr { void (*error_exit)(j_common_ptr); };
struct jpeg_memory_mgr { JSAMPARRAY (*alloc_sarray)(j_common_ptr, int, JDIMENSION, JDIMENSION); };
struct jpeg_source_mgr {
  const unsigned char* next_input_byte;
  size_t bytes_in_buffer;
  void (*init_source)(j_decompress_ptr);
  boolean (*fill_input_buffer)(j_decompress_ptr);
  void (*skip_input_data)(j_decompress_ptr, long);
  boolean (*resync_to_restart)(j_decompress_ptr, int);
  void (*term_source)(j_decompress_ptr);
};
struct jpeg_common_struct { struct jpeg_error_mgr* err; struct jpeg_memory_mgr* mem; };
struct jpeg_decompress_struct {
  struct jpeg_error_mgr* err;
  struct jpeg_memory_mgr* mem;
  struct jpeg_source_mgr* src;
  JDIMENSION output_width, output_height;
};
static inline void efr_no_jpeg(void) { fprintf(stderr, "oracle/host_on_cpu/jpeglib.h: no JPEG decoder in this build\n"); abort(); }
static inline struct jpeg_error_mgr* jpeg_std_error(struct jpeg_error_mgr* e) { return e; }
static inline void jpeg_create_decompress(j_decompress_ptr) { efr_no_jpeg(); }
static inline boolean jpeg_resync_to_restart(j_decompress_ptr, int) { efr_no_jpeg(); return 0; }
static inline int jpeg_read_header(j_decompress_ptr, boolean) { efr_no_jpeg(); return 0; }
static inline void jpeg_calc_output_dimensions(j_decompress_ptr) { efr_no_jpeg(); }
static inline boolean jpeg_start_decompress(j_decompress_ptr) { efr_no_jpeg(); return 0; }
static inline JDIMENSION jpeg_read_scanlines(j_decompress_ptr, JSAMPARRAY, JDIMENSION) { efr_no_jpeg(); return 0; }
static inline boolean jpeg_finish_decompress(j_decompress_ptr) { efr_no_jpeg(); return 0; }
static inline void jpeg_destroy_decompress(j_decompress_ptr) { efr_no_jpeg(); }
#endif
