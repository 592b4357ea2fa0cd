// flow_px4.hpp -- OpticalFlowPX4: sparse 8x8 SAD block matching (published
// PX4Flow configuration) behind the calcFlow() API, computed on MI355X.
#pragma once

#include "optical_flow.hpp"

#define DEFAULT_SEARCH_SIZE 4              // +-4 px (BASELINE.json configs[0])
#define DEFAULT_FLOW_FEATURE_THRESHOLD 30  // 4x4 gradient gate
#define DEFAULT_FLOW_VALUE_THRESHOLD 3000  // SAD acceptance gate

class OpticalFlowPX4 : public OpticalFlow {
public:
	OpticalFlowPX4(float f_length_x, float f_length_y, int ouput_rate = DEFAULT_OUTPUT_RATE,
		       int img_width = DEFAULT_IMAGE_WIDTH, int img_height = DEFAULT_IMAGE_HEIGHT,
		       int search_size = DEFAULT_SEARCH_SIZE,
		       int flow_feature_threshold = DEFAULT_FLOW_FEATURE_THRESHOLD,
		       int flow_value_threshold = DEFAULT_FLOW_VALUE_THRESHOLD);
	~OpticalFlowPX4();

	int calcFlow(uint8_t *img_current, const uint32_t &img_time_us, int &dt_us, float &flow_x,
		     float &flow_y);
};
