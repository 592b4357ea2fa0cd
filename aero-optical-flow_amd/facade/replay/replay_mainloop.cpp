// replay_mainloop -- demonstrates the drop-in boundary without OpenCV/MAVLink.
//
// Replays the engine-facing steps of Mainloop::camera_callback()
// (/root/reference/src/mainloop.cpp:278-331) against the facade with the same
// argument types: centre crop to getImageWidth() x getImageHeight() (:295-298),
// timestamps relative to the first frame (:305-311), contiguous copy of the
// region of interest (:317-320), calcFlow (:322), release, negative-return gate
// (:327-331), and prints what would go into OPTICAL_FLOW_RAD (:359-371).
// Frames come from a raw file: <n frames> of camera_width x camera_height bytes.
//
//   replay_mainloop frames.raw cam_w cam_h crop_w crop_h fps [rate] [fx fy]
#include <flow_opencv.hpp>  // the header the reference includes (mainloop.h:36)
#include <optical_flow_rad.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
	if (argc < 7) {
		std::fprintf(stderr, "usage: %s frames.raw cam_w cam_h crop_w crop_h fps [rate] [fx fy]\n", argv[0]);
		return 2;
	}
	const uint32_t camera_width = std::atoi(argv[2]), camera_height = std::atoi(argv[3]);
	const uint32_t crop_width = std::atoi(argv[4]), crop_height = std::atoi(argv[5]);
	const double fps = std::atof(argv[6]);
	const int flow_output_rate = argc > 7 ? std::atoi(argv[7]) : DEFAULT_OUTPUT_RATE;
	const float focal_length_x = argc > 9 ? (float)std::atof(argv[8]) : 216.6677f;  // main.cpp:60
	const float focal_length_y = argc > 9 ? (float)std::atof(argv[9]) : 216.2457f;  // main.cpp:61

	// mainloop.cpp:423-424
	OpticalFlowOpenCV *_optical_flow = new OpticalFlowOpenCV(focal_length_x, focal_length_y,
								 flow_output_rate, crop_width, crop_height);
	if (!_optical_flow) return 1;
	std::printf("# engine: %s  DEFAULT_OUTPUT_RATE=%u\n", _optical_flow->lastError(), DEFAULT_OUTPUT_RATE);

	FILE *f = std::fopen(argv[1], "rb");
	if (!f) { std::perror(argv[1]); return 1; }
	std::vector<uint8_t> frame((size_t)camera_width * camera_height);
	uint64_t camera_initial_timestamp = 0;
	bool have_initial = false;
	uint8_t tx_seq = 1;  // the reference sends one COMMAND_LONG first (mavlink_tcp.cpp:74-76)
	for (int k = 0; std::fread(frame.data(), 1, frame.size(), f) == frame.size(); k++) {
		int dt_us = 0;
		float flow_x_ang = 0, flow_y_ang = 0;
		uint64_t img_time_us = 1000000ull + (uint64_t)(k * 1.0e6 / fps);  // "camera" clock
		// crop (mainloop.cpp:295-298)
		const int cx = camera_width / 2 - _optical_flow->getImageWidth() / 2;
		const int cy = camera_height / 2 - _optical_flow->getImageHeight() / 2;
		const int cw = _optical_flow->getImageWidth(), ch = _optical_flow->getImageHeight();
		// relative timestamps (mainloop.cpp:305-311)
		if (have_initial) {
			img_time_us -= camera_initial_timestamp;
		} else {
			camera_initial_timestamp = img_time_us;
			have_initial = true;
			img_time_us = 0;
		}
		// contiguous copy of the ROI (mainloop.cpp:317-320), released right after the call
		uint8_t *cropped = (uint8_t *)std::malloc((size_t)cw * ch);
		for (int y = 0; y < ch; y++)
			std::memcpy(cropped + (size_t)y * cw, frame.data() + (size_t)(cy + y) * camera_width + cx, cw);
		int flow_quality = _optical_flow->calcFlow(cropped, (uint32_t)img_time_us, dt_us, flow_x_ang, flow_y_ang);
		std::memset(cropped, 0xAA, (size_t)cw * ch);  // the engine must not keep the pointer
		std::free(cropped);
		if (flow_quality < 0) {  // mainloop.cpp:327-331
			std::printf("%d skip\n", k);
			continue;
		}
		// what mainloop.cpp:359-371 puts on the wire
		std::printf("%d quality=%d integration_time_us=%d integrated_x=%.9g integrated_y=%.9g", k,
			    flow_quality, dt_us, flow_x_ang, flow_y_ang);
		// the frame mavlink_tcp.cpp:142-162 would send (no gyro source here: zeros)
		OpticalFlowRad msg;
		fillOpticalFlowRad(msg, 5000000ull, img_time_us, dt_us, flow_x_ang, flow_y_ang, 0.0, 0.0, 0.0, flow_quality);
		uint8_t wire[OPTICAL_FLOW_RAD_MAX_FRAME];
		const size_t n = packOpticalFlowRad(msg, tx_seq++, MAVLINK_SYSTEM_ID_DEFAULT, MAVLINK_COMPONENT_ID_CAMERA, wire);
		std::printf(" mavlink=");
		for (size_t b = 0; b < n; b++) std::printf("%02x", wire[b]);
		std::printf("\n");
	}
	std::fclose(f);
	delete _optical_flow;  // mainloop.cpp:456
	return 0;
}
