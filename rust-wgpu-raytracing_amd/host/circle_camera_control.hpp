// Host-side mirror of the camera-controller surface.
//   trait CameraController          /root/reference/src/camera_control.rs:4-7
//   CircleCameraController          /root/reference/src/circle_camera_control.rs:7-106
// winit's WindowEvent is replaced by a plain {key, pressed} record: there is no
// window on a headless GPU node, events come from a script or a test.
#pragma once

#include "camera.hpp"

namespace rwr {

enum class VirtualKeyCode { Space, LShift, W, A, S, D, Up, Left, Down, Right, Other };
enum class ElementState { Pressed, Released };

// The subset of winit::event::WindowEvent::KeyboardInput the controller reads
// (circle_camera_control.rs:33-43).
struct KeyboardInput {
    ElementState state = ElementState::Pressed;
    VirtualKeyCode virtual_keycode = VirtualKeyCode::Other;
};

struct CameraController {
    virtual ~CameraController() = default;
    virtual bool process_events(const KeyboardInput &event) = 0;
    virtual void update_camera(Camera &camera) const = 0;
};

class CircleCameraController : public CameraController {
public:
    explicit CircleCameraController(float speed) : speed_(speed) {}

    // circle_camera_control.rs:31-74
    bool process_events(const KeyboardInput &event) override
    {
        const bool is_pressed = event.state == ElementState::Pressed;
        switch (event.virtual_keycode) {
            case VirtualKeyCode::Space: is_up_pressed_ = is_pressed; return true;
            case VirtualKeyCode::LShift: is_down_pressed_ = is_pressed; return true;
            case VirtualKeyCode::W:
            case VirtualKeyCode::Up: is_forward_pressed_ = is_pressed; return true;
            case VirtualKeyCode::A:
            case VirtualKeyCode::Left: is_left_pressed_ = is_pressed; return true;
            case VirtualKeyCode::S:
            case VirtualKeyCode::Down: is_backward_pressed_ = is_pressed; return true;
            case VirtualKeyCode::D:
            case VirtualKeyCode::Right: is_right_pressed_ = is_pressed; return true;
            default: return false;
        }
    }

    // circle_camera_control.rs:76-105.  Space/LShift are recorded above but, as
    // in the reference, never used here.
    void update_camera(Camera &camera) const override
    {
        Vector3 forward = camera.target - camera.eye;
        const Vector3 forward_norm = forward.normalize();
        float forward_mag = forward.magnitude();

        if (is_forward_pressed_ && forward_mag > speed_) camera.eye += forward_norm * speed_;
        if (is_backward_pressed_) camera.eye -= forward_norm * speed_;

        const Vector3 right = forward_norm.cross(camera.up);

        forward = camera.target - camera.eye;
        forward_mag = forward.magnitude();

        if (is_right_pressed_) camera.eye = camera.target - (forward + right * speed_).normalize() * forward_mag;
        if (is_left_pressed_) camera.eye = camera.target - (forward - right * speed_).normalize() * forward_mag;
    }

    void set_pressed_mask(uint32_t keys)
    {
        is_forward_pressed_ = keys & RWR_KEY_FORWARD;
        is_backward_pressed_ = keys & RWR_KEY_BACKWARD;
        is_left_pressed_ = keys & RWR_KEY_LEFT;
        is_right_pressed_ = keys & RWR_KEY_RIGHT;
        is_up_pressed_ = keys & RWR_KEY_UP;
        is_down_pressed_ = keys & RWR_KEY_DOWN;
    }

private:
    float speed_;
    bool is_up_pressed_ = false;
    bool is_down_pressed_ = false;
    bool is_forward_pressed_ = false;
    bool is_backward_pressed_ = false;
    bool is_left_pressed_ = false;
    bool is_right_pressed_ = false;
};

}  // namespace rwr
