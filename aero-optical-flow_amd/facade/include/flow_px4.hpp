// flow_px4.hpp -- OpticalFlowPX4: sparse 8x8 SAD block matching (published
// PX4Flow configuration) behind the calcFlow() API, computed on MI355X.
#pragma once

#include "optical_flow.hpp"

#define DEFAULT_SEARCH_SIZE 4              // +-4 px (BASELINE.json configs[0])
#define DEFAULT_FLOW_FEATURE_THRESHOLD 30  // 4x4 gradient gate
#define DEFAULT_FLOW_VALUE_THRESHOLD 3000  // SAD acceptance gate

// One tile of the sparse grid as a tracked feature: where its centre was in the previous
// image and where the SAD match (with half-pixel refinement) puts it in the current one.
struct TrackedFeature {
	float prev_x, prev_y;  // tile centre in the previous image (px)
	float cur_x, cur_y;    // matched position in the current image (px, half-pixel steps)
	int sad;               // SAD of the best integer match
	bool accepted;         // passed the gradient gate and the SAD gate (votes for the flow)
};

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

	// Sparse feature tracking between two explicit images (BASELINE.json's north_star
	// names a trackFeatures entry next to calcFlow; the reference never calls it, so
	// the signature is this build's).  Stateless: does not touch the frame kept by
	// calcFlow.  Writes up to `capacity` tiles in grid order and returns the number of
	// tiles of the grid, or a negative value when the engine failed.
	int trackFeatures(const uint8_t *img_prev, const uint8_t *img_current, TrackedFeature *features,
			  int capacity);
};
