// flow_opencv.hpp -- OpticalFlowOpenCV, the class the mounted reference
// instantiates (/root/reference/src/mainloop.h:36,77; mainloop.cpp:423-424).
//
// Same constructor arguments and calcFlow() contract as the call sites need.
// The engine underneath is NOT OpenCV feature tracking: per BASELINE.json's
// north_star this build serves the API with the SAD block-matching path on
// MI355X (num_feat picks the sparse grid: ceil(sqrt(num_feat)) tiles per axis;
// conf_multi is accepted for source compatibility and unused).  No OpenCV
// headers or libraries are needed to compile against or link this class.
//
// What ships: 8x8 tiles, +-4 px search with half-pixel refinement on TWO levels
// (half resolution first, its match predicts the full-resolution search) with the
// frame means equalised per level: per-frame flow up to +-9.5 px (the half-resolution
// match must lie within its own +-4 px), beyond which the result is not meaningful.  Frames whose size cannot carry two levels
// (odd width or height, or too small for the half-resolution grid) run one level:
// +-4.5 px.  getPyramidLevels() / setSearchPyramid() report and change it.
// Quality is NOT a feature count: it is accepted_tiles * 255 / tiles of the grid
// (a tile is accepted when it passes the 4x4 gradient gate and its best SAD is below
// the SAD gate), 0 when ten tiles or fewer were accepted; while integrating towards
// the output rate, the mean quality of the frames whose quality was > 0.
#pragma once

#include "optical_flow.hpp"

#define DEFAULT_NUMBER_OF_FEATURES 20
#define DEFAULT_CONFIDENCE_MULTIPLIER 1.645f

class OpticalFlowOpenCV : public OpticalFlow {
public:
	OpticalFlowOpenCV(float f_length_x, float f_length_y, int ouput_rate = DEFAULT_OUTPUT_RATE,
			  int img_width = DEFAULT_IMAGE_WIDTH, int img_height = DEFAULT_IMAGE_HEIGHT,
			  int num_feat = DEFAULT_NUMBER_OF_FEATURES,
			  float conf_multi = DEFAULT_CONFIDENCE_MULTIPLIER);
	~OpticalFlowOpenCV();

	int calcFlow(uint8_t *img_current, const uint32_t &img_time_us, int &dt_us, float &flow_x,
		     float &flow_y);

private:
	int num_features;
	float confidence_multiplier;
};
