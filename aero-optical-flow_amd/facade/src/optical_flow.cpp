// Facade over the C ABI (include/aof.h): previous-frame keeping lives in the
// engine's streaming entry point; rate limiting and the pixel -> angle
// conversion live here, as the calcFlow() contract at
// /root/reference/src/mainloop.cpp:322-331,359-363 requires.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "aof.h"
#include "flow_opencv.hpp"
#include "flow_px4.hpp"

OpticalFlow::OpticalFlow(float f_length_x, float f_length_y, int ouput_rate, int img_width,
			 int img_height)
	: image_width(img_width), image_height(img_height), focal_length_x(f_length_x),
	  focal_length_y(f_length_y), output_rate(ouput_rate), time_last_pub(0), _ctx(NULL)
{
	std::snprintf(_err, sizeof(_err), "engine not opened");
	initLimitRate();
}

OpticalFlow::~OpticalFlow()
{
	if (_ctx) aof_destroy(_ctx);
}

const char *OpticalFlow::lastError() const { return _err; }

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

int OpticalFlow::integrate(const uint8_t *img, uint32_t img_time_us, int &dt_us, float &flow_x,
			   float &flow_y)
{
	bool first = false;
	float px = 0.0f, py = 0.0f;
	int flow_quality = pixelFlow(img, &px, &py, &first);
	if (first) return 0;  // nothing to compare the very first frame with
	flow_quality = limitRate(flow_quality, img_time_us, &dt_us, &px, &py);
	if (flow_quality < 0) return flow_quality;
	flow_x = std::atan2(px, focal_length_x);  // pixel flow -> angular flow (rad)
	flow_y = std::atan2(py, focal_length_y);
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
	openEngine(&p);
}

OpticalFlowOpenCV::~OpticalFlowOpenCV() {}

int OpticalFlowOpenCV::calcFlow(uint8_t *img_current, const uint32_t &img_time_us, int &dt_us,
				float &flow_x, float &flow_y)
{
	return integrate(img_current, img_time_us, dt_us, flow_x, flow_y);
}
