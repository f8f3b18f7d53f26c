"""one env's state as the `State` message of the reference's gRPC agent service
(mujoco_mpc/mjpc/grpc/agent.proto:75-83), checked against the protobuf runtime with the message built from the
same field numbers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def state_message_class():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name = "hb_test_agent_state.proto"
    fdp.package = "hbtest"
    fdp.syntax = "proto2"
    msg = fdp.message_type.add()
    msg.name = "State"
    f = msg.field.add(); f.name = "time"; f.number = 1; f.type = f.TYPE_DOUBLE; f.label = f.LABEL_OPTIONAL
    for k, name in enumerate(["qpos", "qvel", "act", "mocap_pos", "mocap_quat", "userdata"]):
        f = msg.field.add(); f.name = name; f.number = 2 + k; f.type = f.TYPE_DOUBLE; f.label = f.LABEL_REPEATED
        f.options.packed = True
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    desc = pool.FindMessageTypeByName("hbtest.State")
    if hasattr(message_factory, "GetMessageClass"):
        return message_factory.GetMessageClass(desc)
    return message_factory.MessageFactory(pool).GetPrototype(desc)


def test_state_message_round_trip(hbmod, humanoid_model, gpu):
    State = state_message_class()
    m = humanoid_model
    b = hbmod.Batch(m, 6, gpu)
    b.reset(perturb=True)
    b.rollout_halton(40)
    st = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    # ours -> protobuf runtime
    for e in (0, 5):
        msg = State()
        msg.ParseFromString(b.state_to_proto(e))
        assert msg.time == pytest.approx(st[e, 0]) and len(msg.qpos) == m.nq and len(msg.qvel) == m.nv
        assert np.array_equal(np.array(msg.qpos), st[e, 1:1 + m.nq]) and np.array_equal(np.array(msg.qvel), st[e, 1 + m.nq:1 + m.nq + m.nv])
        assert len(msg.act) == 0 and len(msg.userdata) == 0
    # protobuf runtime -> ours: env 2 takes env 5's state; warm start is cleared, other envs untouched
    msg = State()
    msg.time = float(st[5, 0]); msg.qpos.extend(st[5, 1:1 + m.nq]); msg.qvel.extend(st[5, 1 + m.nq:1 + m.nq + m.nv])
    b.state_from_proto(2, msg.SerializeToString())
    st2 = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    assert np.array_equal(st2[2, :1 + m.nq + m.nv], st[5, :1 + m.nq + m.nv]) and not st2[2, 1 + m.nq + m.nv:].any()
    assert np.array_equal(np.delete(st2, 2, axis=0), np.delete(st, 2, axis=0))
    # a message with only qvel leaves time and qpos alone
    only = State(); only.qvel.extend(np.zeros(m.nv))
    b.state_from_proto(2, only.SerializeToString())
    st3 = b.get_state(hbmod.STATE_INTEGRATION, dtype=np.float64)
    assert np.array_equal(st3[2, :1 + m.nq], st2[2, :1 + m.nq]) and not st3[2, 1 + m.nq:].any()
    # refused: wrong length, activations (this engine has none), truncated bytes
    bad = State(); bad.qpos.extend(np.zeros(m.nq - 1))
    with pytest.raises(hbmod.HbError):
        b.state_from_proto(0, bad.SerializeToString())
    act = State(); act.act.extend([0.5])
    with pytest.raises(hbmod.HbError):
        b.state_from_proto(0, act.SerializeToString())
    with pytest.raises(hbmod.HbError):
        b.state_from_proto(0, msg.SerializeToString()[:-3])
    with pytest.raises(hbmod.HbError):
        b.state_to_proto(99)
