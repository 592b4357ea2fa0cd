// Facade over the C ABI (include/aof.h): previous-frame keeping lives in the
// engine's streaming entry point; rate limiting and the pixel -> angle
// conversion live here, as the calcFlow() contract at
// /root/reference/src/mainloop.cpp:322-331,359-363 requires.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "aof.h"
#include "aof_math.h"
#include "flow_opencv.hpp"
#include "flow_px4.hpp"

OpticalFlow::OpticalFlow(float f_length_x, float f_length_y, int ouput_rate, int img_width,
			 int img_height)
	: image_width(img_width), image_height(img_height), focal_length_x(f_length_x),
	  focal_length_y(f_length_y), output_rate(ouput_rate), time_last_pub(0), _ctx(NULL), _resident(false)
{
	std::snprintf(_err, sizeof(_err), "engine not opened");
	initLimitRate();
}

OpticalFlow::~OpticalFlow()
{
	if (_ctx) aof_destroy(_ctx);
}

const char *OpticalFlow::lastError() const { return _err; }

bool OpticalFlow::engineOk() const { return _ctx != NULL; }

bool OpticalFlow::openEngine(const void *params)
{
	const aof_params *p = static_cast<const aof_params *>(params);
	int rc = aof_create(p, 0, &_ctx);
	if (rc) {
		// Constructors cannot fail in the reference's protocol (mainloop.cpp:423-428) and
		// nothing may throw: stay alive, never publish, and say why once.
		_ctx = NULL;
		std::snprintf(_err, sizeof(_err), "aof_create failed: %s", aof_strerror(rc));
		std::fprintf(stderr, "OpticalFlow: %s (no CPU fallback; flow output disabled)\n", _err);
		return false;
	}
	std::snprintf(_err, sizeof(_err), "ok");
	return true;
}

bool OpticalFlow::setSearchPyramid(int levels, bool mean_subtract)
{
	if (!_ctx) return false;
	aof_params p;
	if (aof_get_params(_ctx, &p)) return false;
	p.pyramid_levels = levels;
	p.mean_subtract = mean_subtract ? 1 : 0;
	if (aof_params_check(&p)) return false;
	aof_ctx *fresh = NULL;
	if (aof_create(&p, 0, &fresh)) return false;
	aof_destroy(_ctx);
	_ctx = fresh;
	if (_resident) aof_set_stream_resident(_ctx, 1);
	initLimitRate();
	return true;
}

bool OpticalFlow::setResidentKernel(bool on)
{
	if (!_ctx) return false;
	if (aof_set_stream_resident(_ctx, on ? 1 : 0)) return false;
	_resident = on;
	return true;
}

int OpticalFlow::getPyramidLevels() const
{
	aof_params p;
	if (!_ctx || aof_get_params(_ctx, &p)) return 0;
	return p.pyramid_levels;
}

void OpticalFlow::initLimitRate()
{
	sum_flow_x = 0.0f;
	sum_flow_y = 0.0f;
	sum_flow_quality = 0;
	valid_frame_count = 0;
}

int OpticalFlow::limitRate(int flow_quality, const uint32_t frame_time_us, int *dt_us, float *flow_x,
			   float *flow_y)
{
	if (output_rate <= 0) {  // no limit: publish every frame
		*dt_us = (int)(frame_time_us - time_last_pub);
		time_last_pub = frame_time_us;
		return flow_quality;
	}
	if (flow_quality > 0) {
		sum_flow_x += *flow_x;
		sum_flow_y += *flow_y;
		sum_flow_quality += flow_quality;
		valid_frame_count++;
	}
	if ((float)(frame_time_us - time_last_pub) > 1.0e6f / (float)output_rate) {
		int average_flow_quality = 0;
		if (valid_frame_count > 0)
			average_flow_quality =
				(int)std::floor((float)sum_flow_quality / (float)valid_frame_count);
		*flow_x = sum_flow_x;
		*flow_y = sum_flow_y;
		initLimitRate();
		*dt_us = (int)(frame_time_us - time_last_pub);
		time_last_pub = frame_time_us;
		return average_flow_quality;
	}
	return -1;  // still integrating: the caller skips this frame (mainloop.cpp:327-331)
}

int OpticalFlow::pixelFlow(const uint8_t *img, float *flow_x, float *flow_y, bool *first)
{
	*flow_x = *flow_y = 0.0f;
	*first = false;
	if (!_ctx || !img) return 0;
	aof_flow f;
	int rc = aof_stream_push_host(_ctx, img, &f);
	if (rc < 0) {
		std::snprintf(_err, sizeof(_err), "%s", aof_last_error(_ctx));
		return 0;
	}
	if (rc == 1) {
		*first = true;
		return 0;
	}
	*flow_x = f.flow_x;
	*flow_y = f.flow_y;
	return f.quality;
}

int OpticalFlow::blockMatches(const uint8_t *img_prev, const uint8_t *img_current, void *blocks,
			      uint8_t *subdirs, int capacity, int *grid, int *tile, int *value_threshold)
{
	if (!_ctx || !img_prev || !img_current) return -1;
	aof_params p;
	aof_get_params(_ctx, &p);
	int32_t g[6];
	if (aof_grid(&p, 0, &g[0], &g[1], &g[2], &g[3], &g[4], &g[5])) return -1;
	const int n = g[4] * g[5];
	if (capacity < n) return -1;
	aof_flow f;
	int rc = aof_flow_pair_host(_ctx, img_prev, img_current, static_cast<aof_block *>(blocks), subdirs, &f);
	if (rc < 0) {
		std::snprintf(_err, sizeof(_err), "%s", aof_last_error(_ctx));
		return -1;
	}
	for (int k = 0; k < 6; k++) grid[k] = g[k];
	*tile = p.tile;
	*value_threshold = p.value_threshold > 0xFFFF ? 0xFFFF : p.value_threshold;
	return n;
}

int OpticalFlow::gridTiles() const
{
	if (!_ctx) return -1;
	aof_params p;
	aof_get_params(_ctx, &p);
	int32_t nx = 0, ny = 0;
	if (aof_grid(&p, 0, NULL, NULL, NULL, NULL, &nx, &ny)) return -1;
	return nx * ny;
}

int OpticalFlow::integrate(const uint8_t *img, uint32_t img_time_us, int &dt_us, float &flow_x,
			   float &flow_y)
{
	if (!engineOk()) return -1;  // no engine (no gfx950 device): never publish, see lastError()
	bool first = false;
	float px = 0.0f, py = 0.0f;
	int flow_quality = pixelFlow(img, &px, &py, &first);
	if (first) return 0;  // nothing to compare the very first frame with
	flow_quality = limitRate(flow_quality, img_time_us, &dt_us, &px, &py);
	if (flow_quality < 0) return flow_quality;
	// pixel flow -> angular flow (rad): atan2 as a fixed sequence of IEEE operations (include/aof_math.h), the
	// same bits the device pipeline (aof_sequence_device) produces
	flow_x = aof_atan2f(px, focal_length_x);
	flow_y = aof_atan2f(py, focal_length_y);
	return flow_quality;
}

// ---- OpticalFlowPX4 -----------------------------------------------------------

OpticalFlowPX4::OpticalFlowPX4(float f_length_x, float f_length_y, int ouput_rate, int img_width,
			       int img_height, int search_size, int flow_feature_threshold,
			       int flow_value_threshold)
	: OpticalFlow(f_length_x, f_length_y, ouput_rate, img_width, img_height)
{
	aof_params p;
	aof_params_px4flow(&p, img_width, img_height, search_size, flow_feature_threshold,
			   flow_value_threshold);
	openEngine(&p);
}

OpticalFlowPX4::~OpticalFlowPX4() {}

int OpticalFlowPX4::calcFlow(uint8_t *img_current, const uint32_t &img_time_us, int &dt_us,
			     float &flow_x, float &flow_y)
{
	return integrate(img_current, img_time_us, dt_us, flow_x, flow_y);
}

int OpticalFlowPX4::trackFeatures(const uint8_t *img_prev, const uint8_t *img_current,
				  TrackedFeature *features, int capacity)
{
	static const int half_x[9] = {1, 1, 0, -1, -1, -1, 0, 1, 0};  // half-pixel direction -> x step
	static const int half_y[9] = {0, 1, 1, 1, 0, -1, -1, -1, 0};
	int grid[6], tile = 0, vthr = 0;
	const int tiles = gridTiles();
	if (tiles <= 0) return -1;
	std::vector<aof_block> blocks((size_t)tiles);
	std::vector<uint8_t> subdirs((size_t)tiles, 8);
	int n = blockMatches(img_prev, img_current, blocks.data(), subdirs.data(), tiles, grid, &tile, &vthr);
	if (n < 0) return n;
	for (int k = 0; k < n && k < capacity; k++) {
		const int bx = k % grid[4], by = k / grid[4];
		const aof_block &b = blocks[k];
		const int sd = subdirs[k] <= 8 ? subdirs[k] : 8;
		TrackedFeature &t = features[k];
		t.prev_x = (float)(grid[0] + bx * grid[2]) + tile * 0.5f;
		t.prev_y = (float)(grid[1] + by * grid[3]) + tile * 0.5f;
		const bool searched = b.sad != AOF_SAD_SKIPPED;
		t.accepted = searched && (int)b.sad < vthr;
		t.sad = searched ? (int)b.sad : -1;
		t.cur_x = t.prev_x + (searched ? (float)b.dx + 0.5f * (float)half_x[sd] : 0.0f);
		t.cur_y = t.prev_y + (searched ? (float)b.dy + 0.5f * (float)half_y[sd] : 0.0f);
	}
	return n;
}

// ---- OpticalFlowOpenCV ----------------------------------------------------------

OpticalFlowOpenCV::OpticalFlowOpenCV(float f_length_x, float f_length_y, int ouput_rate,
				     int img_width, int img_height, int num_feat, float conf_multi)
	: OpticalFlow(f_length_x, f_length_y, ouput_rate, img_width, img_height),
	  num_features(num_feat), confidence_multiplier(conf_multi)
{
	aof_params p;
	aof_params_px4flow(&p, img_width, img_height, DEFAULT_SEARCH_SIZE,
			   DEFAULT_FLOW_FEATURE_THRESHOLD, DEFAULT_FLOW_VALUE_THRESHOLD);
	int per_axis = 1;
	while (per_axis * per_axis < num_feat) per_axis++;
	p.num_blocks = per_axis;
	// The class mainloop.cpp:423 creates runs at 128x128 and ~75 Hz on a moving vehicle: a
	// single +-4 search would pin fast motion at the search limit while still reporting a
	// plausible quality.  Two levels with per-level mean equalisation reach +-9.5 px and
	// shrug off the auto-exposure steps; geometries that cannot carry a half-resolution grid
	// keep the single level (getPyramidLevels() says which).
	aof_params two = p;
	two.pyramid_levels = 2;
	two.mean_subtract = 1;
	openEngine(aof_params_check(&two) == 0 ? &two : &p);
}

OpticalFlowOpenCV::~OpticalFlowOpenCV() {}

int OpticalFlowOpenCV::calcFlow(uint8_t *img_current, const uint32_t &img_time_us, int &dt_us,
				float &flow_x, float &flow_y)
{
	return integrate(img_current, img_time_us, dt_us, flow_x, flow_y);
}
