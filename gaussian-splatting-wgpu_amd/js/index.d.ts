// Type declarations for the Node host (the reference is TypeScript; no tsc is available offline,
// so these are hand-written and mirror src/renderer.ts, src/camera.ts, src/ply.ts).
export type Mat4 = Float32Array;
export type Vec3 = Float32Array;
export interface CanvasLike { width: number; height: number; manual?: boolean; onFrame?(rgba: Uint8Array, width: number, height: number): void; }
export interface CameraRaw { id: number; img_name: string; width: number; height: number; position: number[]; rotation: number[][]; fx: number; fy: number; }
export class Camera {
  height: number; width: number; viewMatrix: Mat4; perspective: Mat4; focalX: number; focalY: number; scaleModifier: number;
  constructor(height: number, width: number, viewMatrix: Mat4, perspective: Mat4, focalX: number, focalY: number, scaleModifier: number);
  static default(canvas?: CanvasLike): Camera;
  setScale(scale: number): void; setFocalX(f: number): void; setFocalY(f: number): void;
  getPosition(): Vec3; getProjMatrix(): Mat4;
  translate(x: number, y: number, z: number): void; rotate(x: number, y: number, z: number): void;
  packUniforms(canvasWidth: number, canvasHeight: number, out?: Float32Array): Float32Array;
}
export class InteractiveCamera {
  constructor(camera: Camera, canvas: CanvasLike);
  static default(canvas: CanvasLike): InteractiveCamera;
  key(k: string): boolean; drag(movementX: number, movementY: number): void; wheel(deltaY: number): void;
  setNewCamera(c: Camera): void; isDirty(): boolean; getCamera(): Camera;
}
export class PackedGaussians {
  numGaussians: number; sphericalHarmonicsDegree: number; readonly nShCoeffs: number;
  gaussianLayout: { size: number }; gaussianArrayLayout: { size: number }; gaussiansBuffer: ArrayBuffer;
  constructor(arrayBuffer: ArrayBuffer);
  static fromRecords(arrayBuffer: ArrayBuffer, numGaussians: number): PackedGaussians;
  static fromFile(path: string): PackedGaussians;
}
export class Renderer {
  canvas: CanvasLike; interactiveCamera: InteractiveCamera; numGaussians: number; tileSize: number; numIntersections: number; numFrames: number;
  constructor(canvas: CanvasLike, interactiveCamera: InteractiveCamera, device: number | { ordinal: number; flags?: number; shareWith?: Renderer }, gaussians: PackedGaussians, tileSize: number);
  animate(): Promise<void>; destroy(): Promise<void>;
  renderUniforms(uniforms: Float32Array, debug?: boolean): void; readPixels(): Uint8Array; readBuffer(which: number): ArrayBuffer;
  stats(): { numGaussians: number; numVisible: number; numIntersections: number; numProcessed: number; numTiles: number; sortPasses: number; frames: number; frameUs: number; stageUs: number[]; numEvaluated: number; depthOrdered: number; tightBinning: number; graphFrames: number; capacity: number; maxIntersectionsSeen: number; truncatedFrames: number };
}
export function loadFileAsArrayBuffer(path: string): Promise<ArrayBuffer>;
export function cameraFromJSON(raw: CameraRaw, canvasW: number, canvasH: number): Camera;
export function loadCameraFile(path: string, canvas?: CanvasLike): { name: string; camera: Camera }[];
export function getProjectionMatrix(znear: number, zfar: number, fovX: number, fovY: number): Mat4;
export function focal2fov(focal: number, pixels: number): number;
export function writePPM(file: string, rgba: Uint8Array, width: number, height: number): void;
