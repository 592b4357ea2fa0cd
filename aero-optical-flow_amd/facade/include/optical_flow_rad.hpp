// optical_flow_rad.hpp -- OPTICAL_FLOW_RAD producer.
//
// What Mainloop::camera_callback() does with the engine's outputs
// (/root/reference/src/mainloop.cpp:359-371: field mapping, gyro axis swap, constants) and
// what Mavlink_TCP::optical_flow_rad_msg_write() puts on the wire
// (/root/reference/src/mavlink_tcp.cpp:142-162: mavlink_msg_optical_flow_rad_encode +
// mavlink_msg_to_send_buffer, system id 1, component id MAV_COMP_ID_CAMERA,
// src/mavlink_tcp.h:66-67).  The reference's serializer (modules/mavlink_c) is an empty
// directory in the mount, so the byte layout follows the public MAVLink 2 serialization
// rules for message 106 of the common dialect; like the flow engine it is unpinned against
// reference code.  Host-side C++11, no dependencies.
#pragma once

#include <cstddef>
#include <cstdint>

struct OpticalFlowRad {  // field-for-field mavlink_optical_flow_rad_t
	uint64_t time_usec;
	uint32_t integration_time_us;
	float integrated_x, integrated_y;
	float integrated_xgyro, integrated_ygyro, integrated_zgyro;
	uint32_t time_delta_distance_us;
	float distance;
	int16_t temperature;
	uint8_t sensor_id;
	uint8_t quality;
};

enum {
	OPTICAL_FLOW_RAD_MSG_ID = 106,
	OPTICAL_FLOW_RAD_PAYLOAD_LEN = 44,
	OPTICAL_FLOW_RAD_CRC_EXTRA = 138,
	OPTICAL_FLOW_RAD_MAX_FRAME = 10 + 44 + 2,
	MAVLINK_SYSTEM_ID_DEFAULT = 1,   // mavlink_tcp.h:66
	MAVLINK_COMPONENT_ID_CAMERA = 100  // MAV_COMP_ID_CAMERA, mavlink_tcp.h:67
};

// mainloop.cpp:359-371: fills the message from calcFlow's outputs and the integrated gyro
// ("switch to match pixel directions": xgyro = -gyro_y, ygyro = gyro_x).
void fillOpticalFlowRad(OpticalFlowRad &msg, uint64_t offset_timestamp_usec, uint64_t img_time_us,
			int dt_us, float flow_x_ang, float flow_y_ang, double gyro_x, double gyro_y,
			double gyro_z, int flow_quality);

// MAVLink 2 frame: 0xFD, len, incompat, compat, seq, sysid, compid, msgid[3], payload with
// trailing zero bytes truncated (at least 1 byte), CRC-16/MCRF4XX over everything after the
// start byte plus the message's CRC_EXTRA.  Returns the frame length (<= 56).
size_t packOpticalFlowRad(const OpticalFlowRad &msg, uint8_t seq, uint8_t system_id,
			  uint8_t component_id, uint8_t out[OPTICAL_FLOW_RAD_MAX_FRAME]);

uint16_t mavlinkCrcAccumulate(const uint8_t *data, size_t len, uint16_t crc /* 0xFFFF to start */);
