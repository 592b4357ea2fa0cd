// Plain-C handles on the facade classes, so that non-C++ harnesses (the Python
// tests via ctypes) drive the real C++ classes rather than a re-implementation.
// Nothing throws across the C boundary: the classes are made with new (std::nothrow) -- a null handle
// tells the caller, as /root/reference/src/mainloop.cpp:425-428 checks its own `new` -- and the one array
// a delegate needs likewise.
#include <new>

#include "flow_opencv.hpp"
#include "flow_px4.hpp"
#include "optical_flow_rad.hpp"

extern "C" {

void *aof_facade_px4_create(float fx, float fy, int output_rate, int w, int h, int search,
			    int feature_threshold, int value_threshold)
{
	return new (std::nothrow) OpticalFlowPX4(fx, fy, output_rate, w, h, search, feature_threshold,
						 value_threshold);
}

void *aof_facade_opencv_create(float fx, float fy, int output_rate, int w, int h)
{
	// exactly the five arguments /root/reference/src/mainloop.cpp:423-424 passes
	return new (std::nothrow) OpticalFlowOpenCV(fx, fy, output_rate, w, h);
}

void aof_facade_destroy(void *flow) { delete static_cast<OpticalFlow *>(flow); }

int aof_facade_calc_flow(void *flow, uint8_t *img, uint32_t t_us, int *dt_us, float *flow_x,
			 float *flow_y)
{
	return static_cast<OpticalFlow *>(flow)->calcFlow(img, t_us, *dt_us, *flow_x, *flow_y);
}

int aof_facade_px4_track_features(void *flow, const uint8_t *prev, const uint8_t *cur, float *out6, int capacity)
{
	// out6: capacity rows of {prev_x, prev_y, cur_x, cur_y, sad, accepted}
	OpticalFlowPX4 *px4 = static_cast<OpticalFlowPX4 *>(flow);
	TrackedFeature *tmp = new (std::nothrow) TrackedFeature[capacity > 0 ? capacity : 1];
	if (!tmp) return -1;
	int n = px4->trackFeatures(prev, cur, tmp, capacity);
	for (int k = 0; k < n && k < capacity; k++) {
		out6[6 * k + 0] = tmp[k].prev_x; out6[6 * k + 1] = tmp[k].prev_y;
		out6[6 * k + 2] = tmp[k].cur_x;  out6[6 * k + 3] = tmp[k].cur_y;
		out6[6 * k + 4] = (float)tmp[k].sad; out6[6 * k + 5] = tmp[k].accepted ? 1.0f : 0.0f;
	}
	delete[] tmp;
	return n;
}

int aof_facade_pack_optical_flow_rad(uint64_t offset_ts, uint64_t img_time_us, int dt_us, float fx, float fy,
				     double gyro_x, double gyro_y, double gyro_z, int quality, uint8_t seq,
				     uint8_t *out56)
{
	OpticalFlowRad m;
	fillOpticalFlowRad(m, offset_ts, img_time_us, dt_us, fx, fy, gyro_x, gyro_y, gyro_z, quality);
	return (int)packOpticalFlowRad(m, seq, MAVLINK_SYSTEM_ID_DEFAULT, MAVLINK_COMPONENT_ID_CAMERA, out56);
}

unsigned aof_facade_mavlink_crc(const uint8_t *data, int len) { return mavlinkCrcAccumulate(data, (size_t)len, 0xFFFF); }

int aof_facade_set_search_pyramid(void *flow, int levels, int mean_subtract)
{
	return static_cast<OpticalFlow *>(flow)->setSearchPyramid(levels, mean_subtract != 0) ? 1 : 0;
}
int aof_facade_pyramid_levels(void *flow) { return static_cast<OpticalFlow *>(flow)->getPyramidLevels(); }
int aof_facade_set_resident(void *flow, int on) { return static_cast<OpticalFlow *>(flow)->setResidentKernel(on != 0) ? 1 : 0; }

int aof_facade_image_width(void *flow) { return static_cast<OpticalFlow *>(flow)->getImageWidth(); }
int aof_facade_image_height(void *flow) { return static_cast<OpticalFlow *>(flow)->getImageHeight(); }
const char *aof_facade_last_error(void *flow) { return static_cast<OpticalFlow *>(flow)->lastError(); }
int aof_facade_default_output_rate(void) { return DEFAULT_OUTPUT_RATE; }

}  // extern "C"
