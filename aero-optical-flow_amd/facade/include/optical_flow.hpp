// optical_flow.hpp -- base class of the drop-in flow engine facade.
//
// Drop-in for the class API the reference consumes from its PX4 OpticalFlow
// submodule (absent from /root/reference; signatures are pinned by the call
// sites only): constructor /root/reference/src/mainloop.cpp:423-424, getters
// :295-297, calcFlow :322, negative-return gate :327-331, macro
// DEFAULT_OUTPUT_RATE /root/reference/src/main.cpp:145.  C++11, no OpenCV, no
// exceptions; all pixel work runs in the gfx950 kernels behind include/aof.h.
#pragma once

#include <cstdint>

#define DEFAULT_OUTPUT_RATE 15
#define DEFAULT_IMAGE_WIDTH 64
#define DEFAULT_IMAGE_HEIGHT 64

struct aof_ctx;  // C ABI context (include/aof.h)

class OpticalFlow {
public:
	virtual ~OpticalFlow();

	// img: image_width x image_height 8-bit grey, row stride == width; the caller may
	// release it as soon as the call returns (mainloop.cpp:317-324).
	// Returns < 0 while integrating towards the output rate (outputs untouched), else
	// the quality 0..255 with dt_us and the angular flow (rad) filled in.
	virtual int calcFlow(uint8_t *img_current, const uint32_t &img_time_us, int &dt_us,
			     float &flow_x, float &flow_y) = 0;

	inline void setImageWidth(int img_width) { image_width = img_width; }
	inline void setImageHeight(int img_height) { image_height = img_height; }
	inline void setFocalLengthX(float f_length) { focal_length_x = f_length; }
	inline void setFocalLengthY(float f_length) { focal_length_y = f_length; }
	inline void setOutputRate(int out_rate) { output_rate = out_rate; }
	inline int getImageWidth() { return image_width; }
	inline int getImageHeight() { return image_height; }
	inline float getFocalLengthX() { return focal_length_x; }
	inline float getFocalLengthY() { return focal_length_y; }
	inline int getOutputRate() { return output_rate; }

	// Search reach of the engine.  One level: +-(search + 0.5) px per frame (4.5 px with the
	// defaults); two levels add a half-resolution pass whose match (itself within +-search
	// half-resolution pixels) predicts the full-resolution search: +-(2 * search + 1.5) px
	// (9.5 px); mean_subtract equalises the frame means per
	// level first (the exposure steps of the reference's auto-exposure loop,
	// /root/reference/src/mainloop.cpp:197-275).  Re-creates the engine: the frame kept by
	// calcFlow() and the running integration are dropped.  False (and the previous engine
	// stays) when the geometry does not allow it (two levels need even width and height and a
	// grid that still fits the half-resolution frame).
	bool setSearchPyramid(int levels, bool mean_subtract);
	int getPyramidLevels() const;

	// Opt-in: serve calcFlow() from a kernel that STAYS on the device and takes each frame through a
	// mailbox in pinned memory, instead of one kernel launch per call (aof_set_stream_resident in aof.h:
	// about 10 us per call instead of 25-35 us; the kernel leaves by itself after 50 ms without a
	// frame and is started again by the next call).  Same results bit for bit.
	bool setResidentKernel(bool on);

	// Text of the last engine error ("ok" when healthy); never throws.
	const char *lastError() const;
	// False when the GPU engine could not be created (no gfx950 device).  There is no CPU
	// fallback: calcFlow() then returns -1 for every frame, i.e. nothing is ever published.
	bool engineOk() const;

protected:
	OpticalFlow(float f_length_x, float f_length_y, int ouput_rate, int img_width, int img_height);

	void initLimitRate();
	int limitRate(int flow_quality, const uint32_t frame_time_us, int *dt_us, float *flow_x,
		      float *flow_y);
	// Pushes one frame through the engine; returns quality (0 for the very first
	// frame or on engine failure) and the pixel flow against the previous frame.
	int pixelFlow(const uint8_t *img, float *flow_x, float *flow_y, bool *first);
	bool openEngine(const void *params);  // aof_params
	int gridTiles() const;  // tiles of the level-0 grid, -1 without an engine
	// Block records of one explicit image pair (no streaming state involved).
	int blockMatches(const uint8_t *img_prev, const uint8_t *img_current, void *blocks,
			 uint8_t *subdirs, int capacity, int *grid /* x0,y0,step_x,step_y,nx,ny */,
			 int *tile, int *value_threshold);
	// pixelFlow + limitRate + pixel -> angle: the whole calcFlow() contract.
	int integrate(const uint8_t *img, uint32_t img_time_us, int &dt_us, float &flow_x,
		      float &flow_y);

	int image_width;
	int image_height;
	float focal_length_x;  // px
	float focal_length_y;
	int output_rate;       // Hz; <= 0 publishes every frame
	float sum_flow_x;
	float sum_flow_y;
	int sum_flow_quality;
	int valid_frame_count;
	uint32_t time_last_pub;

private:
	OpticalFlow(const OpticalFlow &);
	OpticalFlow &operator=(const OpticalFlow &);
	aof_ctx *_ctx;
	bool _resident;  // setResidentKernel(): carried over when the engine is re-created
	char _err[160];
};
