"""ROS wire structs of MCPTAM's map exchange, so that maps travelling between the reference's client and server nodes
(`ModifyMap` service, src/NetworkManager.cc) can be replayed through the HIP back end.

Messages (field order and types as in /root/reference/msg/*.msg):

* ``NetworkMapPoint``      msg/NetworkMapPoint.msg        (filled by NetworkManager::MapPoint_To_AddMsg, src/NetworkManager.cc:985-1013)
* ``NetworkMeasurement``   msg/NetworkMeasurement.msg     (KeyFrame_To_AddMsg / AddMsg_To_KeyFrame, :966-982)
* ``NetworkKeyFrame``      msg/NetworkKeyFrame.msg
* ``NetworkMultiKeyFrame`` msg/NetworkMultiKeyFrame.msg   (MultiKeyFrame_To_AddMsg, :590-611)
* ``NetworkOutlier``       msg/NetworkOutlier.msg

Encoding = ROS 1 message serialisation (little endian; ``string`` = uint32 length + bytes; ``T[]`` = uint32 count +
elements; ``T[n]`` = n elements, no count; ``bool`` = one byte; nested messages inline; ``sensor_msgs/Image`` = Header
{uint32 seq, time stamp (2 x uint32), string frame_id}, uint32 height, uint32 width, string encoding, uint8
is_bigendian, uint32 step, uint8[] data).  TooN objects travel as the text their ``operator<<`` writes to a
``std::stringstream`` with the stream's default 6 significant digits (``Vector<3>``: the three numbers each followed by a
blank; ``SE3``: three rows "r0 r1 r2 t" [3P-memory: TooN/se3.h, TooN/TooN.h]); `precision` reproduces that (6) or keeps
the doubles exactly (17).

``map_from_messages`` / ``messages_from_map`` convert between a list of these messages and map_io.MapFile, from where
``map_io.problem_from_map`` leads to ``Problem.populate(ChainBundle)``.
"""
import struct
from dataclasses import dataclass, field, fields

import numpy as np

from . import map_io


# ---------------------------------------------------------------------------------------------------- messages
@dataclass
class Header:                       # std_msgs/Header
    seq: int = 0
    stamp_secs: int = 0
    stamp_nsecs: int = 0
    frame_id: str = ""
    _spec = (("seq", "uint32"), ("stamp_secs", "uint32"), ("stamp_nsecs", "uint32"), ("frame_id", "string"))


@dataclass
class Image:                        # sensor_msgs/Image
    header: Header = field(default_factory=Header)
    height: int = 0
    width: int = 0
    encoding: str = ""
    is_bigendian: int = 0
    step: int = 0
    data: bytes = b""
    _spec = (("header", Header), ("height", "uint32"), ("width", "uint32"), ("encoding", "string"), ("is_bigendian", "uint8"),
             ("step", "uint32"), ("data", "uint8[]"))

    @staticmethod
    def from_array(a):
        """mono8 image message of a (h, w) uint8 array."""
        a = np.ascontiguousarray(a, dtype=np.uint8)
        return Image(height=a.shape[0], width=a.shape[1], encoding="mono8", step=a.shape[1], data=a.tobytes())

    def to_array(self):
        if self.height == 0 or self.width == 0:
            return np.zeros((0, 0), dtype=np.uint8)
        a = np.frombuffer(self.data, dtype=np.uint8).reshape(self.height, self.step)
        return a[:, :self.width].copy()


@dataclass
class NetworkMeasurement:
    nLevel: int = 0
    bSubPix: bool = False
    v2RootPos: tuple = (0.0, 0.0)
    eSource: int = 0
    mapPointId: str = ""
    _spec = (("nLevel", "uint8"), ("bSubPix", "bool"), ("v2RootPos", "float64[2]"), ("eSource", "uint8"), ("mapPointId", "string"))


@dataclass
class NetworkKeyFrame:
    mse3CamFromBase: str = ""
    mse3CamFromWorld: str = ""
    image: Image = field(default_factory=Image)
    mask: Image = field(default_factory=Image)
    mvMeasurements: list = field(default_factory=list)
    mdSceneDepthMean: float = 0.0
    mdSceneDepthSigma: float = 0.0
    mCamName: str = ""
    mParentId: str = ""
    _spec = (("mse3CamFromBase", "string"), ("mse3CamFromWorld", "string"), ("image", Image), ("mask", Image),
             ("mvMeasurements", [NetworkMeasurement]), ("mdSceneDepthMean", "float64"), ("mdSceneDepthSigma", "float64"),
             ("mCamName", "string"), ("mParentId", "string"))


@dataclass
class NetworkMultiKeyFrame:
    mse3BaseFromWorld: str = ""
    mvKeyFrames: list = field(default_factory=list)
    mbFixed: bool = False
    mdTotalDepthMean: float = 0.0
    mId: str = ""
    _spec = (("mse3BaseFromWorld", "string"), ("mvKeyFrames", [NetworkKeyFrame]), ("mbFixed", "bool"), ("mdTotalDepthMean", "float64"),
             ("mId", "string"))


@dataclass
class NetworkMapPoint:
    mv3WorldPos: str = ""
    mnSourceLevel: int = 0
    mv3PixelRight_W: str = ""
    mv3PixelDown_W: str = ""
    mirCenter: tuple = (0.0, 0.0)
    mId: str = ""
    mSourceId: str = ""
    mSourceCamName: str = ""
    mbFixed: bool = False
    mbOptimized: bool = False
    _spec = (("mv3WorldPos", "string"), ("mnSourceLevel", "uint8"), ("mv3PixelRight_W", "string"), ("mv3PixelDown_W", "string"),
             ("mirCenter", "float64[2]"), ("mId", "string"), ("mSourceId", "string"), ("mSourceCamName", "string"),
             ("mbFixed", "bool"), ("mbOptimized", "bool"))


@dataclass
class NetworkOutlier:
    mMKFId: str = ""
    mCamName: str = ""
    mapPointId: str = ""
    _spec = (("mMKFId", "string"), ("mCamName", "string"), ("mapPointId", "string"))


# ---------------------------------------------------------------------------------------------------- ROS 1 serialisation
_SCALAR = {"uint8": "<B", "bool": "<B", "uint32": "<I", "int32": "<i", "float64": "<d"}


def _put(out, typ, v):
    if isinstance(typ, list):                                   # variable-length array of messages
        out.append(struct.pack("<I", len(v)))
        for e in v:
            _put(out, typ[0], e)
    elif not isinstance(typ, str):                              # nested message
        for name, t in typ._spec:
            _put(out, t, getattr(v, name))
    elif typ == "string":
        b = v.encode("utf-8") if isinstance(v, str) else bytes(v)
        out.append(struct.pack("<I", len(b)))
        out.append(b)
    elif typ == "uint8[]":
        b = bytes(v)
        out.append(struct.pack("<I", len(b)))
        out.append(b)
    elif typ.endswith("]"):                                     # fixed-size array "T[n]"
        base, n = typ[:-1].split("[")
        if len(v) != int(n):
            raise ValueError("field of type %s needs %s elements" % (typ, n))
        for e in v:
            out.append(struct.pack(_SCALAR[base], e))
    else:
        out.append(struct.pack(_SCALAR[typ], int(v) if typ != "float64" else float(v)))


def serialize(msg):
    """ROS 1 wire bytes of a message object."""
    out = []
    _put(out, type(msg), msg)
    return b"".join(out)


def _get(buf, pos, typ):
    if isinstance(typ, list):
        (n,), pos = struct.unpack_from("<I", buf, pos), pos + 4
        items = []
        for _ in range(n):
            e, pos = _get(buf, pos, typ[0])
            items.append(e)
        return items, pos
    if not isinstance(typ, str):
        kw = {}
        for name, t in typ._spec:
            kw[name], pos = _get(buf, pos, t)
        return typ(**kw), pos
    if typ in ("string", "uint8[]"):
        (n,), pos = struct.unpack_from("<I", buf, pos), pos + 4
        if pos + n > len(buf):
            raise ValueError("truncated message")
        raw = bytes(buf[pos:pos + n])
        return (raw.decode("utf-8") if typ == "string" else raw), pos + n
    if typ.endswith("]"):
        base, n = typ[:-1].split("[")
        vals = []
        for _ in range(int(n)):
            (v,), pos = struct.unpack_from(_SCALAR[base], buf, pos), pos + struct.calcsize(_SCALAR[base])
            vals.append(v)
        return tuple(vals), pos
    (v,), pos = struct.unpack_from(_SCALAR[typ], buf, pos), pos + struct.calcsize(_SCALAR[typ])
    return (bool(v) if typ == "bool" else v), pos


def deserialize(cls, buf):
    """Message object of class `cls` from ROS 1 wire bytes (the whole buffer must be consumed)."""
    try:
        msg, pos = _get(buf, 0, cls)
    except struct.error as exc:
        raise ValueError("truncated message: %s" % exc)
    if pos != len(buf):
        raise ValueError("%d trailing bytes after %s" % (len(buf) - pos, cls.__name__))
    return msg


# ---------------------------------------------------------------------------------------------------- TooN text fields
def vector_text(v, precision=6):
    """TooN ``operator<<(ostream&, Vector)``: every element followed by one blank."""
    return "".join(("%%.%dg " % precision) % float(x) for x in v)


def se3_text(R, t, precision=6):
    """TooN ``operator<<(ostream&, SE3)``: three lines, rotation row then the translation component."""
    f = "%%.%dg" % precision
    return "".join(" ".join(f % float(x) for x in R[i]) + " " + (f % float(t[i])) + "\n" for i in range(3))


def parse_vector(s, n=3):
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if v.size != n:
        raise ValueError("expected %d numbers, got %r" % (n, s))
    return v


def parse_se3(s):
    """``istream >> SE3``: twelve numbers, row-wise [R | t]; the rotation is re-orthonormalised as TooN's ``SO3::coerce`` does
    (Gram-Schmidt on the rows [3P-memory])."""
    m = parse_vector(s, 12).reshape(3, 4)
    R, t = m[:, :3].copy(), m[:, 3].copy()
    R[0] /= np.linalg.norm(R[0])
    R[1] -= R[0] * (R[0] @ R[1])
    R[1] /= np.linalg.norm(R[1])
    R[2] -= R[0] * (R[0] @ R[2])
    R[2] -= R[1] * (R[1] @ R[2])
    R[2] /= np.linalg.norm(R[2])
    return R, t


# ---------------------------------------------------------------------------------------------------- MapFile <-> messages
def messages_from_map(m, precision=6, images=None):
    """(list of NetworkMultiKeyFrame, list of NetworkMapPoint) carrying the map `m` (map_io.MapFile) the way the client sends
    a whole map: one ADD per MKF with its KeyFrames and their measurements, one ADD per point.  Ids are the decimal MKF /
    point numbers.  `images`: optional dict (mkf, camera name) -> (h, w) uint8 array."""
    C = len(m.cam_names)
    cam_from_base = [map_io._inverse(map_io.matrix_from_quat(m.cam_quat[c]), m.cam_pos[c]) for c in range(C)]
    by_kf = {}
    for j in range(len(m.ms_pt)):
        by_kf.setdefault((int(m.ms_mkf[j]), m.ms_cam[j]), []).append(j)
    mkfs = []
    for k in range(len(m.mkf_pos)):
        bR, bt = map_io._inverse(map_io.matrix_from_quat(m.mkf_quat[k]), m.mkf_pos[k])
        msg = NetworkMultiKeyFrame(mse3BaseFromWorld=se3_text(bR, bt, precision), mbFixed=(k == 0), mId=str(k))
        for c, name in enumerate(m.cam_names):
            cR, ct = cam_from_base[c]
            wR, wt = cR @ bR, cR @ bt + ct
            kf = NetworkKeyFrame(mse3CamFromBase=se3_text(cR, ct, precision), mse3CamFromWorld=se3_text(wR, wt, precision),
                                 mCamName=name, mParentId=str(k))
            if images and (k, name) in images:
                kf.image = Image.from_array(images[(k, name)])
            for j in by_kf.get((k, name), []):
                lvl = int(round(np.log(max(float(m.ms_noise[j]), 1.0)) / np.log(4.0)))
                kf.mvMeasurements.append(NetworkMeasurement(nLevel=lvl, bSubPix=False, v2RootPos=(float(m.ms_uv[j, 0]), float(m.ms_uv[j, 1])),
                                                            eSource=0, mapPointId=str(int(m.ms_pt[j]))))
            msg.mvKeyFrames.append(kf)
        mkfs.append(msg)
    pts = [NetworkMapPoint(mv3WorldPos=vector_text(m.pt_world[i], precision), mv3PixelRight_W=vector_text(np.zeros(3), precision),
                           mv3PixelDown_W=vector_text(np.zeros(3), precision), mId=str(i), mSourceId=str(int(m.pt_parent_mkf[i])),
                           mSourceCamName=m.pt_parent_cam[i]) for i in range(len(m.pt_world))]
    return mkfs, pts


def map_from_messages(mkfs, pts):
    """map_io.MapFile of a set of ADD messages.  MKFs and points are numbered in list order; measurements of points that
    are not in `pts` are dropped (the server does the same for ids it does not know, src/NetworkManager.cc:930-940)."""
    if not mkfs:
        raise ValueError("no MultiKeyFrame message")
    mkf_no = {msg.mId: k for k, msg in enumerate(mkfs)}
    pt_no = {msg.mId: i for i, msg in enumerate(pts)}
    names = [kf.mCamName for kf in mkfs[0].mvKeyFrames]
    cp, cq = np.zeros((len(names), 3)), np.zeros((len(names), 4))
    for c, kf in enumerate(mkfs[0].mvKeyFrames):                 # one pose per camera name, from the first MKF (BundleAdjusterMulti.cc:94-106)
        R, t = parse_se3(kf.mse3CamFromBase)
        Ri, ti = map_io._inverse(R, t)
        cp[c], cq[c] = ti, map_io.quat_from_matrix(Ri)
    kp, kq = np.zeros((len(mkfs), 3)), np.zeros((len(mkfs), 4))
    ms = []
    for k, msg in enumerate(mkfs):
        R, t = parse_se3(msg.mse3BaseFromWorld)
        Ri, ti = map_io._inverse(R, t)
        kp[k], kq[k] = ti, map_io.quat_from_matrix(Ri)
        for kf in msg.mvKeyFrames:
            for meas in kf.mvMeasurements:
                if meas.mapPointId in pt_no:
                    ms.append((k, kf.mCamName, pt_no[meas.mapPointId], meas.v2RootPos[0], meas.v2RootPos[1], 4.0 ** meas.nLevel))
    world = np.array([parse_vector(p.mv3WorldPos) for p in pts], dtype=np.float64).reshape(len(pts), 3)
    for p in pts:
        if p.mSourceId not in mkf_no:
            raise ValueError("point %s refers to unknown MultiKeyFrame %s" % (p.mId, p.mSourceId))
    return map_io.MapFile(cam_names=names, cam_pos=cp, cam_quat=cq, mkf_pos=kp, mkf_quat=kq, pt_world=world,
                          pt_parent_mkf=np.array([mkf_no[p.mSourceId] for p in pts], dtype=np.int32),
                          pt_parent_cam=[p.mSourceCamName for p in pts],
                          ms_mkf=np.array([r[0] for r in ms], dtype=np.int32), ms_cam=[r[1] for r in ms],
                          ms_pt=np.array([r[2] for r in ms], dtype=np.int32),
                          ms_uv=np.array([[r[3], r[4]] for r in ms], dtype=np.float64).reshape(len(ms), 2),
                          ms_noise=np.array([r[5] for r in ms], dtype=np.float64))


def outliers_to_messages(outliers, mkf_ids, cam_names, point_ids):
    """NetworkOutlier messages for ChainBundle::GetOutlierMeasurements() tuples already translated to (MKF number, camera
    index, point number), as BundleAdjusterMulti hands them to the map maker (src/BundleAdjusterMulti.cc:297-331)."""
    return [NetworkOutlier(mMKFId=str(mkf_ids[k]), mCamName=cam_names[c], mapPointId=str(point_ids[i])) for k, c, i in outliers]
