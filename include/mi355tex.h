/* mi355tex.h — image ingest for the texture array.
 *
 * The reference turns each encoded image a glTF carries into one 1024x1024 RGBA8 layer with
 *     createImageBitmap(new Blob([data]), {resizeWidth: 1024, resizeHeight: 1024})
 *     copyExternalImageToTexture({source: bmp}, {texture, origin: [0, 0, i]}, [1024, 1024])
 * (/root/reference/src/renderer/ResourceManager.ts:153-198; white 1024x1024 fallback when decoding fails, :200-208).
 * Here the decode runs on the host (this library, plain C ABI, no GPU) and the resize on the GPU
 * (rt_upload_texture_image in mi355rt.h).
 *
 * Decoders are written from the format specifications (PNG: ISO/IEC 15948 + RFC 1950/1951; JPEG: ITU-T T.81 Huffman-coded
 * sequential and progressive DCT, 8-bit, with JFIF YCbCr -> RGB).  What the browser does beyond the specifications
 * is not pinned by anything in the reference; the choices made here are stated in DESIGN.md §4.5:
 *   - 16-bit PNG samples keep their high byte; gAMA / iCCP / sRGB chunks are ignored (the shader does no sRGB decode)
 *   - alpha stays straight (not premultiplied); grey and palette images expand to RGBA
 *   - JPEG: integer "slow" IDCT and triangle ("fancy") chroma upsampling as published with the IJG library,
 *     Adobe APP14 transform flag honoured, CMYK/YCCK refused
 */
#ifndef MI355TEX_H
#define MI355TEX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mt_image {
  uint32_t width, height;
  uint8_t* rgba; /* width * height * 4 bytes, rows top to bottom, owned by the library until mt_free */
} mt_image;

enum { MT_OK = 0, MT_ERR_FORMAT = -1, MT_ERR_UNSUPPORTED = -2, MT_ERR_CORRUPT = -3, MT_ERR_MEMORY = -4 };
enum { MT_KIND_UNKNOWN = 0, MT_KIND_PNG = 1, MT_KIND_JPEG = 2 };

/* Container sniffing by magic bytes (what the browser does with an untyped Blob). */
int mt_probe(const uint8_t* data, size_t size);
/* Decode one encoded image to straight RGBA8.  Returns MT_OK or a negative code (message: mt_last_error). */
int mt_decode(const uint8_t* data, size_t size, mt_image* out);
void mt_free(mt_image* img);
/* Thread-local description of the last failure in this thread. */
const char* mt_last_error(void);
/* RFC 1950 zlib stream -> bytes (exposed for tests); returns the decoded size or a negative code. */
long mt_inflate(const uint8_t* data, size_t size, uint8_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
