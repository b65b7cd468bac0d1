/* pbrt_mi355_host.h — host-side helpers exported by libpbrt_mi355.so beside the core ABI.
 *
 * These restate, in f32 and in the reference's operation order, the reference constructors whose RESULTS
 * cross the C ABI (matrices, filter table, densified spectra).  A Rust host does not need them: it already
 * owns the originals (file:line cited per function) and passes their fields to pm_set_camera / pm_set_film /
 * pm_set_lights / pm_set_materials.  They exist for the C++/Python harnesses of this repository.
 * None of them touches the GPU.
 */
#ifndef PBRT_MI355_HOST_H
#define PBRT_MI355_HOST_H
#include "pbrt_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SquareMatrix<4>::inverse (core/square_matrix.rs:196-262); returns non-zero if singular */
int pmh_mat4_inverse(const float* m16, float* out16);
/* SquareMatrix<4> * SquareMatrix<4> with mul_add accumulation (core/square_matrix.rs:316-332) */
void pmh_mat4_mul(const float* a16, const float* b16, float* out16);
/* Transform::look_at (core/transform.rs:207-223): out_m = camera_from_world, out_inv = world_from_camera */
int pmh_look_at(const float* pos3, const float* look3, const float* up3, float* out_m16, float* out_inv16);
/* CameraTransform::new + PerspectiveCamera::new + ProjectiveCamera::new
 * (camera/base.rs:265-279,342-380; camera/perspective.rs:30-57; core/transform.rs:236-255) */
int pmh_perspective_camera(const float* world_from_camera_m16, const float* world_from_camera_inv16, float fov_deg,
                           const float* screen_window4 /* min.x min.y max.x max.y */, int res_x, int res_y,
                           float lens_radius, float focal_distance, PmCamera* out,
                           float* world_from_render16_or_null, float* render_from_world16_or_null);
/* GaussianFilter::new + FilterSampler::new (imaging/filter.rs:138-216); vals may be NULL to query nx, ny */
int pmh_gaussian_filter_table(float radius_x, float radius_y, float sigma, float* vals, int cap, int* nx, int* ny);
float pmh_gaussian(float x, float mean, float sd);   /* math.rs:10-13 */
float pmh_fast_exp(float x);                         /* core/routines.rs:47-73 */
/* ILLUMD65 normalised and densified at 360..830 nm (sampling/spectrum.rs:64-65,196-214,269-310) */
void pmh_d65_table(const float* cie_y471, const float* d65_interleaved, int n_interleaved, float* out471);
/* Spectrum::to_photometric of a densely sampled spectrum (sampling/spectrum.rs:152-160) */
float pmh_photometric(const float* cie_y471, const float* table471);
/* BlackbodySpectrum::new(T) densified (sampling/spectrum.rs:376-437); returns the normalisation factor */
float pmh_blackbody_table(float temp_kelvin, float* out471);
/* scale * RGBSigmoidPolynomial(c0,c1,c2) [* illuminant], densified (sampling/spectrum.rs:922-1019) */
void pmh_rgb_spectrum_table(const float* c3, float scale, const float* illuminant471_or_null, float* out471);
float pmh_sigmoid_poly(const float* c3, float lambda);   /* color/color_space.rs:243-280 */
/* RGBColorSpace::to_rgb_coeffs without the (missing) table file: solves the table generator's optimisation
 * problem for one RGB (color/color_space.rs:110-184, util/data.rs:1684-1702) */
int pmh_rgb_to_coeffs(const float* rgb3, const float* cie_x471, const float* cie_y471, const float* cie_z471,
                      const float* illuminant471, const float* xyz_from_rgb9, float* out_c3);
/* The RGBToSpectrumTable the reference loads from src/util/rgbspectrum_srgb.txt (util/data.rs:1684-1702; the file is
 * missing upstream), regenerated: z_nodes64[64] and coeffs[3*64*64*64*3] in CoefficientTable's layout
 * (color/color_space.rs:200-212).  About 786k small optimisations: seconds on a few threads. */
int pmh_rgb_table_generate(const float* cie_x471, const float* cie_y471, const float* cie_z471, const float* illuminant471,
                           const float* xyz_from_rgb9, float* z_nodes64, float* coeffs, int nthreads);
/* RGBToSpectrumTable::convert (color/color_space.rs:128-185): trilinear lookup, f32, the reference's operation order */
void pmh_rgb_table_lookup(const float* z_nodes64, const float* coeffs, const float* rgb3, float* out_c3);
/* RGBColorSpace::new for the sRGB primaries (color/color_space.rs:40-89) */
int pmh_srgb_matrices(const float* cie_x471, const float* cie_y471, const float* cie_z471, const float* illuminant471,
                      float* xyz_from_rgb9, float* rgb_from_xyz9);

/* Image::write_exr (imaging/image.rs:48-57): width x height RGB f32 pixels (row-major, 3 floats per pixel) as a
 * single-part scanline OpenEXR file with FLOAT channels B, G, R, no compression.  Returns non-zero on I/O failure. */
int pmh_write_exr(const char* path, int width, int height, const float* rgb);

#ifdef __cplusplus
}
#endif
#endif
