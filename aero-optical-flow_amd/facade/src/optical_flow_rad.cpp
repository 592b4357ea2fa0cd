#include "optical_flow_rad.hpp"

#include <cstring>

void fillOpticalFlowRad(OpticalFlowRad &msg, uint64_t offset_timestamp_usec, uint64_t img_time_us,
			int dt_us, float flow_x_ang, float flow_y_ang, double gyro_x, double gyro_y,
			double gyro_z, int flow_quality)
{
	msg.time_usec = offset_timestamp_usec + img_time_us;  // mainloop.cpp:360
	msg.integration_time_us = (uint32_t)dt_us;            // :361
	msg.integrated_x = flow_x_ang;                        // :362
	msg.integrated_y = flow_y_ang;                        // :363
	msg.integrated_xgyro = (float)(-gyro_y);              // :364 switch to match pixel directions
	msg.integrated_ygyro = (float)gyro_x;                 // :365
	msg.integrated_zgyro = (float)gyro_z;                 // :366
	msg.time_delta_distance_us = 0;                       // :367
	msg.distance = -1.0f;                                 // :368
	msg.temperature = 0;                                  // :369
	msg.sensor_id = 0;                                    // :370
	msg.quality = (uint8_t)flow_quality;                  // :371
}

uint16_t mavlinkCrcAccumulate(const uint8_t *data, size_t len, uint16_t crc)
{
	for (size_t i = 0; i < len; i++) {
		uint8_t tmp = (uint8_t)(data[i] ^ (uint8_t)(crc & 0xFF));
		tmp = (uint8_t)(tmp ^ (tmp << 4));
		crc = (uint16_t)((crc >> 8) ^ ((uint16_t)tmp << 8) ^ ((uint16_t)tmp << 3) ^ (tmp >> 4));
	}
	return crc;
}

namespace {
template <typename T> void put(uint8_t *&p, T v)
{
	// little-endian wire order; this code runs on little-endian hosts only (x86-64)
	std::memcpy(p, &v, sizeof(T));
	p += sizeof(T);
}
}  // namespace

size_t packOpticalFlowRad(const OpticalFlowRad &m, uint8_t seq, uint8_t system_id, uint8_t component_id,
			  uint8_t out[OPTICAL_FLOW_RAD_MAX_FRAME])
{
	uint8_t payload[OPTICAL_FLOW_RAD_PAYLOAD_LEN];
	uint8_t *p = payload;
	put(p, m.time_usec);            // fields in wire order: by size, then declaration
	put(p, m.integration_time_us);
	put(p, m.integrated_x);
	put(p, m.integrated_y);
	put(p, m.integrated_xgyro);
	put(p, m.integrated_ygyro);
	put(p, m.integrated_zgyro);
	put(p, m.time_delta_distance_us);
	put(p, m.distance);
	put(p, m.temperature);
	put(p, m.sensor_id);
	put(p, m.quality);
	size_t len = OPTICAL_FLOW_RAD_PAYLOAD_LEN;
	while (len > 1 && payload[len - 1] == 0) len--;  // MAVLink 2 payload truncation
	out[0] = 0xFD;
	out[1] = (uint8_t)len;
	out[2] = 0;  // incompat_flags
	out[3] = 0;  // compat_flags
	out[4] = seq;
	out[5] = system_id;
	out[6] = component_id;
	out[7] = (uint8_t)(OPTICAL_FLOW_RAD_MSG_ID & 0xFF);
	out[8] = (uint8_t)((OPTICAL_FLOW_RAD_MSG_ID >> 8) & 0xFF);
	out[9] = (uint8_t)((OPTICAL_FLOW_RAD_MSG_ID >> 16) & 0xFF);
	std::memcpy(out + 10, payload, len);
	uint16_t crc = mavlinkCrcAccumulate(out + 1, 9 + len, 0xFFFF);
	const uint8_t extra = OPTICAL_FLOW_RAD_CRC_EXTRA;
	crc = mavlinkCrcAccumulate(&extra, 1, crc);
	out[10 + len] = (uint8_t)(crc & 0xFF);
	out[11 + len] = (uint8_t)(crc >> 8);
	return 12 + len;
}
